#!/bin/bash
# Run ON THE GPU BOX: one data-parallel mode at C4 against emulated collectives with their footprint (one real rank plays rank 0
# of N, a rank's batch 625k rows), fixed cuts — no sweep: tools/r05_emulate.sh <tag> <ranks:busbw> <mode> <fractions> [bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
tag=$1; r=$2; mode=$3; fr=$4; shift 4
timeout -k 10 300 python3 bench.py --gpus 1 --force-dp --config C4 --rows 1250000 --batch-rows 625000 --emulate-allreduce $r --emulate-load 64 \
    --dp-exchange $mode --upper-fractions $fr --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extra "$@" > gpurun_out/r05/emu_$tag.json 2> gpurun_out/r05/emu_$tag.err
rc=$?
if [ $rc -ne 0 ]; then echo "$tag rc=$rc"; tail -5 gpurun_out/r05/emu_$tag.err; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: stopping"; exit 1; fi
python3 - <<PY
import json
d = [json.loads(l) for l in open("gpurun_out/r05/emu_$tag.json") if l.startswith("{")][-1]
x = d["exchange"]; k = d["kernels"]
print("$tag [$r $mode $fr]", "%.1f G nnz/s" % (d["value"] / 1e9), "ms/step %.4f" % d["ms_per_step"], "exposed", round(x.get("exposed_comm_ms", -1), 3), "busy", round(x.get("comm_busy_ms", -1), 3),
      "kernels/step", {a: round(k[a]["avg_ms"] * 1e3, 1) for a in k}, "sum %.1f" % sum(k[a]["avg_ms"] * 1e3 for a in k))
PY
