#!/bin/bash
# Run ON THE GPU BOX: one tuning key, two values, per configuration (bench.py --tune KEY=VALUE); one bench line each.
#   tools/r03_ab_tune.sh "4=0 4=2" "C3 C2 C5 C4"
cd "$(dirname "$0")/.."
settings=${1:-4=0 4=2}
cfgs=${2:-C3 C2 C5 C4}
for cfg in $cfgs; do
  for kv in $settings; do
    tag=${kv/=/_}
    timeout -k 10 200 python3 bench.py --config $cfg --no-extra --no-cpu-baseline --no-pmc --tune $kv --steps 120 --warmup 12 \
      > gpurun_out/r03_tune_${cfg}_$tag.json 2> gpurun_out/r03_tune_${cfg}_$tag.err || { echo "$cfg $kv FAILED"; tail -3 gpurun_out/r03_tune_${cfg}_$tag.err; exit 1; }
    python3 - <<PY
import json
o = json.load(open("gpurun_out/r03_tune_${cfg}_$tag.json"))
print("$cfg tune $kv: %.2f G nnz/s %.4f ms" % (o["value"] / 1e9, o["ms_per_step"]), {k: round(v["avg_ms"] * 1e3, 1) for k, v in o["kernels"].items()})
PY
  done
done
