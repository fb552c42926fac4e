#!/bin/bash
# rocprofv3 kernel trace of the judged bench command (v5 profile) + tile-order group A/B for gemm256k
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r42
export TMPDIR=/tmp
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r42/trace -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/r42/trace.log 2>&1
tail -1 gpurun_out/r42/trace.log
python tools/rocprof_summarize.py gpurun_out/r42/trace gpurun_out/r42/r01_14B-720p_kernel_trace_summary_v5.json "bench.py 14B-720p, 1 warm-up + 1 timed step, gemm256k" | head -40
cp $(find gpurun_out/r42/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r42/r01_14B-720p_rocprofv3_kernel_stats_v5.csv 2>/dev/null
rm -rf gpurun_out/r42/trace
for grp in 2 8; do echo "== bench_gemm WAN_GEMM_GROUP=$grp"; WAN_GEMM_GROUP=$grp timeout 300 python tools/bench_gemm.py --rounds 3 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' '; echo; done | tee gpurun_out/r42/bench_gemm_group.log
echo "== default"; timeout 300 python tools/bench_gemm.py --rounds 3 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' ' | tee -a gpurun_out/r42/bench_gemm_group.log; echo
