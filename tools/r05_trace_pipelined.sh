#!/bin/bash
# Run ON THE GPU BOX: kernel TIMELINE of the pipelined data-parallel step at C4 against emulated 8 x 300 GB/s collectives
# (rocprofv3 --kernel-trace; the program itself after `--`), condensed by tools/r05_trace_timeline.py into per-step idle gaps
# of the compute stream.   tools/r05_trace_pipelined.sh [mode, default pipelined] [fractions]
cd "$(dirname "$0")/.."
root=$PWD
mode=${1:-pipelined}; fr=${2:-0.04,0.1,0.3}
mkdir -p gpurun_out/r05
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace -d $root/gpurun_out/r05/trace_$mode -o trace --output-format csv -- \
    python3 $root/bench.py --gpus 1 --force-dp --config C4 --rows 1250000 --batch-rows 625000 --emulate-allreduce 8:300 --emulate-load 64 \
    --dp-exchange $mode --upper-fractions $fr --steps 30 --warmup 6 --settle 0.05 --no-cpu-baseline --no-pmc --no-extra > $root/gpurun_out/r05/trace_$mode.log 2>&1)
rc=$?
echo "trace $mode rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: stopping"; exit 1; fi
python3 tools/r05_trace_timeline.py gpurun_out/r05/trace_$mode
