#!/bin/bash
# Run ON THE GPU BOX: what the data-parallel schedule would do at N = 2 and N = 4 against emulated collectives (one real rank plays
# rank 0 of N; the collective's footprint emulated with 32 workgroups), under two batch policies:
#   per-rank batch fixed at 625k rows (the N = 8 line's; round 3/4's bench default at every N), and
#   GLOBAL batch fixed at 5M rows (per-rank batch 5M / N: the same SGD trajectory whatever N).
# Bus bandwidths: 56 % and 84 % of the links' one-way sum ((N-1) x 76.8 GB/s) - the 300 and 450 GB/s of the N = 8 table.
cd "$(dirname "$0")/../.."
run() { # ranks busbw batch_rows rows
  local tag=r04_emulated_dp_c4_$1_$2_b$3
  timeout -k 10 500 python3 bench.py --gpus 1 --force-dp --config C4 --rows $4 --batch-rows $3 --emulate-allreduce $1:$2 --emulate-load 32 \
      --no-cpu-baseline --no-pmc --no-extra > gpurun_out/$tag.json 2> gpurun_out/$tag.err
  local rc=$?
  echo "$tag rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: stopping"; exit 1; fi
}
# (rows = 2 batches at least: a dataset of ONE batch has no dense hot block)
run 2 43 625000 1250000
run 2 43 2500000 5000000
run 2 65 2500000 5000000
run 4 129 625000 1250000
run 4 129 1250000 2500000
run 4 194 1250000 2500000
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r04_emulated_dp_c4_[24]_*_b*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        ex = d.get("exchange", {})
        print(f.split("/")[-1], "step %.3f ms" % d["ms_per_step"], "rank %.2f G nnz/s" % (d["value"] / 1e9), "exposed %.3f ms" % ex.get("exposed_comm_ms", -1), "mode", ex.get("mode"))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable", e)
PY
