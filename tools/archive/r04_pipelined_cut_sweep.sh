#!/bin/bash
cd "$(dirname "$0")/../.." 2>/dev/null
for fr in "0.04,0.1,0.3" "0.035,0.09,0.3" "0.04,0.1,0.25" "0.045,0.11,0.35" "0.04,0.12,0.3" "0.03,0.08,0.25" "0.05,0.1,0.3" "0.04,0.08,0.3" "0.04,0.1,0.2,0.45" "0.04,0.1,0.4"; do
  timeout -k 10 200 python bench.py --gpus 1 --force-dp --config C4 --rows 1250000 --batch-rows 625000 --emulate-allreduce 8:300 --emulate-load 64 \
      --no-cpu-baseline --no-pmc --no-extra --dp-exchange pipelined --upper-fractions $fr > gpurun_out/sweep_one.json 2> gpurun_out/sweep_one.err
  python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/sweep_one.json") if l.startswith("{")][-1])
x = d["exchange"]
print("$fr", "ms/step %.4f" % d["ms_per_step"], "exposed %.3f busy %.3f" % (x.get("exposed_comm_ms", -1), x.get("comm_busy_ms", -1)), {k: round(v["avg_ms"]*1e3) for k, v in d["kernels"].items()})
PY
done
