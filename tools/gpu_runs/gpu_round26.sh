#!/bin/bash
# final profiling pass of the round: kernel trace of one 14B denoise step, HBM traffic counters of the self-attention kernel
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
export TMPDIR=/tmp
echo "== rocprofv3 kernel trace, bench.py 14B-720p (1 warm-up + 1 timed step)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r26_trace -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/r26_trace.log 2>&1
tail -2 gpurun_out/r26_trace.log | cut -c1-600
python tools/rocprof_summarize.py gpurun_out/r26_trace gpurun_out/r26_14B_kernel_trace_summary.json "bench.py 14B-720p, 1 warm-up + 1 timed step" | head -60
cp $(find gpurun_out/r26_trace -name "*kernel_stats.csv" | head -1) gpurun_out/r26_14B_kernel_stats.csv 2>/dev/null
echo "== PMC traffic, 14B self-attention shape (B=2, L=75600, H=40), w64q flat kernel"
for c in FETCH_SIZE WRITE_SIZE; do
 ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/r26_pmc_$c -o a --output-format csv -- python $R/tools/bench_attn.py --variants w64f --rounds 1 --H 40 ) > gpurun_out/r26_pmc_$c.log 2>&1
 python tools/rocprof_summarize.py gpurun_out/r26_pmc_$c gpurun_out/r26_14B_attn_pmc_$c.json "14B self-attention shape, w64q flat, $c" | grep -E "attn_w64q|SIZE"
done
echo "== bench 1.3B-480p"
timeout 600 python bench.py --workload 1.3B-480p --steps 4 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r26_bench_1.3B.json | cut -c1-900
rm -rf gpurun_out/r26_trace gpurun_out/r26_pmc_FETCH_SIZE gpurun_out/r26_pmc_WRITE_SIZE
