#!/bin/bash
# Run ON THE GPU BOX: pages of the dense hot block (fmhip_tune key 12), 4 against 8, per configuration; one bench line each.
set -e
cd "$(dirname "$0")/.."
cfgs=${1:-C3 C5 C4 C2}
pagelist=${2:-4 6 8}
for cfg in $cfgs; do
  for pages in $pagelist; do
    timeout -k 10 200 python3 bench.py --config $cfg --no-extra --no-cpu-baseline --no-pmc --hot-pages $pages --steps 120 --warmup 12 \
      > gpurun_out/r03_pages_${cfg}_$pages.json 2> gpurun_out/r03_pages_${cfg}_$pages.err
    python3 - <<PY
import json
o = json.load(open("gpurun_out/r03_pages_${cfg}_$pages.json"))
print("$cfg pages $pages: %.2f G nnz/s %.4f ms" % (o["value"] / 1e9, o["ms_per_step"]), {k: round(v["avg_ms"] * 1e3, 1) for k, v in o["kernels"].items()},
      "hot feats", o["config"]["dense_hot_block"]["features_backward"], "share left to bwd %.3f" % o["config"]["dense_hot_block"]["share_of_nonzeros_left_to_the_backward"])
PY
  done
done
