#!/bin/bash
# Run ON THE GPU BOX: the data-parallel schedule of C4's shard against emulated collectives (one real rank playing rank 0
# of N; fmhip_comm_emulate / fmhip_comm_emulate_ranks), one bench line per (ranks, bus bandwidth) into gpurun_out/.
#   tools/r03_emulate.sh "8:300 8:200 8:450 4:300 2:300" [extra bench args]
set -e
cd "$(dirname "$0")/.."
specs=${1:-8:300}; shift || true
for spec in $specs; do
  tag=${spec/:/_}
  timeout -k 10 300 python3 bench.py --gpus 1 --force-dp --config C4 --emulate-allreduce $spec --no-cpu-baseline --steps 60 --warmup 8 "$@" \
    > gpurun_out/r03_emulated_dp_c4_$tag.json 2> gpurun_out/r03_emulated_dp_c4_$tag.err
  python3 - <<PY
import json
o = json.load(open("gpurun_out/r03_emulated_dp_c4_$tag.json"))
x = o["exchange"]
best = {}
for t in x.get("cut_tuning", []):
    if t["exchange"] not in best or t["ms_per_step"] < best[t["exchange"]]["ms_per_step"]:
        best[t["exchange"]] = t
print("$spec", "ms/step %.4f" % o["ms_per_step"], "mode", x.get("mode"), "exposed %.3f busy %.3f" % (x.get("exposed_comm_ms", -1), x.get("comm_busy_ms", -1)),
      "plain %.4f" % x["c4_one_gpu"]["ms_per_step"], "split-no-exchange %.4f" % x["per_gpu_without_exchange"]["ms_per_step"],
      "scaling x%.2f of %s" % (x["scaling_vs_c4_one_gpu"] * int("$spec".split(":")[0]), "$spec".split(":")[0]),
      {k: (v["upper_fractions"], round(v["ms_per_step"], 4)) for k, v in best.items()},
      "c3twin", round(x.get("c3_on_every_gpu", {}).get("ms_per_step", -1), 4), x.get("c3_on_every_gpu", {}).get("exchange"))
PY
done
