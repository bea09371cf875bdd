#!/bin/bash
# Run ON THE GPU BOX: the three dense-family exchange modes (dense, sharded, pipelined) at C4 against emulated collectives with
# their footprint — one real rank plays rank 0 of N; cut and mode tuned by the bench itself (exchange.cut_tuning holds every
# candidate).  RATES = "ranks:busbw ..." (default: 8:300 8:450 8:200); a rank's batch is 625k rows.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for r in ${RATES:-8:300 8:450 8:200}; do
  tag=r04_emulated_dp_c4_${r/:/_}_wg64_modes3
  timeout -k 10 500 python bench.py --gpus 1 --force-dp --config C4 --rows 1250000 --batch-rows 625000 --emulate-allreduce $r --emulate-load 64 \
      --no-cpu-baseline --no-pmc --no-extra > gpurun_out/$tag.json 2> gpurun_out/$tag.err
  rc=$?
  echo "$tag rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: stopping"; exit 1; fi
  python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/$tag.json") if l.startswith("{")][-1])
x = d["exchange"]
print("$r", "value %.1f G nnz/s" % (d["value"] / 1e9), "ms/step %.4f" % d["ms_per_step"], "mode", x["mode"], "exposed", round(x.get("exposed_comm_ms", -1), 3), "busy", round(x.get("comm_busy_ms", -1), 3))
best = {}
for t in x["cut_tuning"]:
    k = t["exchange"]
    if k not in best or t["ms_per_step"] < best[k]["ms_per_step"]:
        best[k] = t
for k, t in best.items():
    print("   best", k, t["upper_fractions"], "%.4f ms" % t["ms_per_step"])
PY
done
