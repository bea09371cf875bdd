import json, sys
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.als_fields(0), indent=1))
