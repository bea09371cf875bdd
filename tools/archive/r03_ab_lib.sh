#!/bin/bash
# Run ON THE GPU BOX: the in-tree library against a variant build (tools/build_variant.sh <name> ...), per configuration.
#   tools/r03_ab_lib.sh <name> "C3 C5"
cd "$(dirname "$0")/.."
name=$1
cfgs=${2:-C3 C5}
for cfg in $cfgs; do
  for lib in base $name; do
    if [ $lib = base ]; then unset FMHIP_LIB; else export FMHIP_LIB=$PWD/sparkfm_amd/lib/libfmhip_$lib.so; fi
    timeout -k 10 200 python3 bench.py --config $cfg --no-extra --no-cpu-baseline --no-pmc --steps 120 --warmup 12 \
      > gpurun_out/r03_lib_${cfg}_$lib.json 2> gpurun_out/r03_lib_${cfg}_$lib.err || { echo "$cfg $lib FAILED"; tail -3 gpurun_out/r03_lib_${cfg}_$lib.err; exit 1; }
    python3 - <<PY
import json
o = json.load(open("gpurun_out/r03_lib_${cfg}_$lib.json"))
print("$cfg $lib: %.2f G nnz/s %.4f ms" % (o["value"] / 1e9, o["ms_per_step"]), {k: round(v["avg_ms"] * 1e3, 1) for k, v in o["kernels"].items()}, "mse %.6f" % o["train"]["last_batch_mse"])
PY
  done
done
