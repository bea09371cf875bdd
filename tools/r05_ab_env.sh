#!/bin/bash
# A/B of an environment switch on the GPU box: tools/r05_ab_env.sh VAR "<bench args>" ["<bench args>" ...]
# runs bench.py (headline only) with VAR unset and VAR=1, twice each in alternation, and prints nnz/s, ms/step and kernel times.
var=$1; shift
mkdir -p gpurun_out/r05
for args in "$@"; do
  tag=$(echo "$args" | tr -c 'A-Za-z0-9' '_')
  for rep in 1 2; do
    for on in 0 1; do
      if [ $on = 1 ]; then export $var=1; else unset $var; fi
      python3 bench.py $args --no-extra --no-pmc --no-cpu-baseline > gpurun_out/r05/ab_${var}_${tag}_${on}_${rep}.json 2>/dev/null || echo "FAILED $args $on"
      python3 - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/r05/ab_${var}_${tag}_${on}_${rep}.json") if l.strip()][-1]
print("${var}=%s rep ${rep} [%s]: %.2f G nnz/s, %.4f ms/step, kernels %s" % ("${on}", "${args}", d["value"]/1e9, d["ms_per_step"], {k: round(v["avg_ms"]*1e3,1) for k,v in d["kernels"].items()}))
PY
    done
  done
done
