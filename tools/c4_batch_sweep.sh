#!/bin/bash
# C4 (10M rows x 1M features, k=32) on ONE GPU at the per-rank batch sizes of a FIXED global batch of 5M rows
# (5M / N rows for N = 8, 4, 2, 1): what one rank's compute costs per step when the same job runs on fewer GPUs.
# (rows = 2 batches: a dataset of ONE batch keeps every feature in the sparse streams - no dense hot block - and is another workload)
#   tools/c4_batch_sweep.sh  ->  gpurun_out/c4_b<rows>.log (one bench line each) + a table on stdout
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for b in 625000 1250000 2500000 5000000; do
  python3 bench.py --config C4 --rows $((2 * b)) --batch-rows $b --no-cpu-baseline --no-extra --no-pmc --steps 16 --warmup 4 > gpurun_out/c4_b$b.log 2>&1
done
python3 - <<'PY'
import json
for b in (625000, 1250000, 2500000, 5000000):
    for l in open("gpurun_out/c4_b%d.log" % b):
        if l.startswith("{"):
            d = json.loads(l); k = d["kernels"]
            print(b, round(d["value"] / 1e9, 2), "G nnz/s", round(d["ms_per_step"], 3), "ms", {a: round(k[a]["avg_ms"] * 1e3, 1) for a in k})
PY
