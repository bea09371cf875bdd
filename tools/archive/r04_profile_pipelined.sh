#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the pipelined data-parallel step at C4's width against emulated 8 x 300 GB/s
# collectives (one real rank; cuts (0.04, 0.1, 0.3)) -> gpurun_out/r04_pipelined_kernel_stats.csv, condensed into
# gpurun_out/r04_pipelined_summary.txt (what profiles/r04_pipelined_summary.txt is a copy of).
root=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
mkdir -p $root/gpurun_out
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_pipe -o pipe --output-format csv -- \
  python3 $root/bench.py --gpus 1 --force-dp --config C4 --rows 1250000 --batch-rows 625000 --emulate-allreduce 8:300 --emulate-load 64 \
  --no-cpu-baseline --no-pmc --no-extra --dp-exchange pipelined --upper-fractions 0.04,0.1,0.3 --steps 100 --warmup 10 \
  > $root/gpurun_out/prof_pipe.log 2>&1
echo "rocprofv3 rc=$?"
f=$(find $root/gpurun_out/prof_pipe -name '*kernel_stats.csv' | head -n 1)
echo "stats: $f"
cd $root && python3 tools/summarize_rocprof.py --stats "$f" > gpurun_out/r04_pipelined_summary.txt 2>&1
head -n 30 gpurun_out/r04_pipelined_summary.txt
grep '^{' gpurun_out/prof_pipe.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench under the profiler: ms/step', d['ms_per_step'], 'mode', d['exchange']['mode'])"
