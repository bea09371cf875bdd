#!/bin/bash
# Run ON THE GPU BOX: batch size with the band-affine backward (bench.py --batch-rows), one bench line each.
cd "$(dirname "$0")/.."
cfgs=${1:-C3}
sizes=${2:-125000 250000 500000 1000000}
for cfg in $cfgs; do
  for b in $sizes; do
    timeout -k 10 200 python3 bench.py --config $cfg --no-extra --no-cpu-baseline --no-pmc --batch-rows $b --steps 96 --warmup 12 \
      > gpurun_out/r03_batch_${cfg}_$b.json 2> gpurun_out/r03_batch_${cfg}_$b.err || { echo "$cfg $b FAILED"; tail -3 gpurun_out/r03_batch_${cfg}_$b.err; exit 1; }
    python3 - <<PY
import json
o = json.load(open("gpurun_out/r03_batch_${cfg}_$b.json"))
bp = o["config"]["backward_band_plan"]
print("$cfg batch $b: %.2f G nnz/s %.4f ms" % (o["value"] / 1e9, o["ms_per_step"]), {k: round(v["avg_ms"] * 1e3, 1) for k, v in o["kernels"].items()}, "affine %.2f" % bp["share_band_affine"])
PY
  done
done
