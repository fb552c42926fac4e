#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -30
echo "== bench 14B"
timeout 1200 python bench.py --workload 14B-720p --steps 2 --warmup 1 2>&1 | tee gpurun_out/bench_14B.log | tail -2
echo "== rocprofv3 kernel trace 14B (1 step)"
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof14b -o r01 --output-format csv -- python $OLDPWD/bench.py --workload 14B-720p --steps 1 --warmup 0 --no-cpu-baseline ) > gpurun_out/rocprof14b.log 2>&1
tail -2 gpurun_out/rocprof14b.log
python tools/rocprof_summarize.py gpurun_out/prof14b gpurun_out/r01_14B_kernel_trace.json "14B-720p 1 step" | tail -40
ls gpurun_out/prof14b/* | head
echo "== rocprofv3 PMC passes (1.3B workload)"
for c in FETCH_SIZE WRITE_SIZE; do
 ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace -d $OLDPWD/gpurun_out/pmc_$c -o r01 --output-format csv -- python $OLDPWD/bench.py --workload 1.3B-480p --steps 1 --warmup 0 --no-cpu-baseline ) > gpurun_out/pmc_$c.log 2>&1
 tail -1 gpurun_out/pmc_$c.log
 python tools/rocprof_summarize.py gpurun_out/pmc_$c gpurun_out/r01_1.3B_pmc_$c.json "1.3B-480p $c" | tail -12
done
# keep only summaries + the stats csv (raw traces are large)
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
