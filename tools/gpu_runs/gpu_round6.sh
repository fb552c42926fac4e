#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== bench 1.3B"
timeout 900 python bench.py --workload 1.3B-480p --steps 3 --warmup 1 2>&1 | tee gpurun_out/bench_1.3B.log | tail -1
echo "== bench 14B"
timeout 1200 python bench.py --workload 14B-720p --steps 2 --warmup 1 2>&1 | tee gpurun_out/bench_14B.log | tail -1
echo "== rocprofv3 kernel trace 14B (1 step)"
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof14b_v2 -o r01 --output-format csv -- python $OLDPWD/bench.py --workload 14B-720p --steps 1 --warmup 0 --no-cpu-baseline ) > gpurun_out/rocprof14b_v2.log 2>&1
python tools/rocprof_summarize.py gpurun_out/prof14b_v2 gpurun_out/r01_14B_kernel_trace_v2.json "14B-720p 1 step (attention v2)" > /dev/null
head -8 gpurun_out/prof14b_v2/r01_kernel_stats.csv | cut -c1-200
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
