#!/bin/bash
# Run ON THE GPU BOX: the data-parallel schedule at C4 against emulated collectives (one real rank plays rank 0 of N), with the
# idle-wait emulation of round 3 and with the collective's footprint (fmhip_comm_emulate_load: WGS workgroups streaming the
# payload through HBM for the collective's duration).  One JSON line per run under gpurun_out/.
# A rank's batch is 625k rows at every N here (the table of DESIGN.md section 7; bench.py's own default is the global batch of 5M rows,
# i.e. 5M / N per rank: tools/r04_emulate_n.sh).
#   tools/r04_emulate.sh            (WGS="0 32 64" by default for 8:300; the other rates with 64)
cd "$(dirname "$0")/../.."
run() { # ranks:busbw wgs
  local tag=r04_emulated_dp_c4_${1/:/_}_wg$2
  timeout -k 10 400 python bench.py --gpus 1 --force-dp --config C4 --rows 1250000 --batch-rows 625000 --emulate-allreduce $1 --emulate-load $2 --no-cpu-baseline --no-pmc --no-extra \
      > gpurun_out/$tag.json 2> gpurun_out/$tag.err
  local rc=$?
  echo "$tag rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: stopping"; exit 1; fi
}
for w in ${WGS:-0 32 64}; do run 8:300 $w; done
run 8:200 64
run 8:450 64
run 4:300 64
run 2:300 64
