#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace statistics plus the PMC passes of the bench workload and of
# the HBM-resident leg, one rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# never together with a trace domain).  Raw CSVs land in gpurun_out/<tag>_*; tools/make_pmc_json.py condenses them.
#   tools/profile_round.sh <tag> [bench args...]      (SKIP_C5=1 / ONLY_C5=1: one of the two workloads only; ONLY_EXTRA=1;
#   C5_ROWS=n; PASS_TIMEOUT=seconds per rocprofv3 run; LEG="tools/pmc_leg.py c4 --steps 8 --warmup 4": profile that instead of bench.py;
#   BASIC_ONLY=1: stats / fetch / write / l2 / sq only)
tag=${1:-r02}; shift || true
cd "$(dirname "$0")/.."
root=$PWD
export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --no-extra --no-pmc --steps 12 --warmup 4 --settle 0.1 $*"
if [ -n "$LEG" ]; then BENCH="python3 $root/$LEG"; fi      # LEG="tools/pmc_leg.py c4 ...": another workload than the bench's own
run() { # name, rocprof args...   (every pass bounded: a counter set the hardware refuses makes rocprofv3 abort and then hang)
  local name=$1; shift
  (cd /tmp && timeout -k 5 ${PASS_TIMEOUT:-150} rocprofv3 "$@" -d $root/gpurun_out/${tag}_$name -o $name --output-format csv -- $BENCH > $root/gpurun_out/${tag}_$name.log 2>&1)
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name TIMED OUT: stopping (no further GPU step after a kill)"; exit 1; fi
  if [ $rc -ne 0 ]; then echo "$name failed (rc $rc)"; grep -m2 -i "error\|exceeds" $root/gpurun_out/${tag}_$name.log; return 1; fi
  echo "$name ok"
}
if [ -z "$ONLY_C5" ]; then
  if [ -z "$ONLY_EXTRA" ]; then       # ONLY_EXTRA=1: the five basic passes exist already (same kernels), add the rest
  run stats --kernel-trace --stats
  run fetch --pmc FETCH_SIZE
  run write --pmc WRITE_SIZE
  run l2 --pmc TCC_HIT_sum TCC_MISS_sum
  run sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
  fi
  if [ -n "$BASIC_ONLY" ]; then exit 0; fi
  # the texture path (the "line-request bound" claim) and the raw memory-side request counters behind FETCH_SIZE
  run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE || true              # two TA counters per pass: more are refused (error 38)
  run ta2 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || true
  run ta3 --pmc TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum || true
  run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || true
  run tcp2 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum || true
  run ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum || true
  run ea2 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_REQ_sum || true
fi
# the HBM-resident leg (C5 width) on its own
if [ -z "$SKIP_C5" ]; then
  BENCH="python3 $root/tools/run_c5_shape.py 24 ${C5_ROWS:-6000000}"
  tag=${tag}_c5
  run stats --kernel-trace --stats
  run fetch --pmc FETCH_SIZE
  run write --pmc WRITE_SIZE
  run l2 --pmc TCC_HIT_sum TCC_MISS_sum
  run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE || true
  run ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum || true
  run ea2 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_REQ_sum || true
fi
