#!/bin/bash
# usage: abrun.sh "<bench args>" lib1 lib2 ...   (lib = base or a variant name)
args="$1"; shift
for v in "$@"; do
  if [ $v = base ]; then unset FMHIP_LIB; else export FMHIP_LIB=sparkfm_amd/lib/libfmhip_$v.so; fi
  python bench.py --no-cpu-baseline --no-extra --no-pmc $args > gpurun_out/ab_$v.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_$v.log") if l.startswith("{")][-1]); k=d["kernels"]
print("$v", round(d["value"]/1e9,2), round(d["ms_per_step"],4), {a:round(k[a]["avg_ms"]*1e3,1) for a in k})
PY
done
