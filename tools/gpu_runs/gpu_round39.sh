#!/bin/bash
# gemm256k (BK=64, five-unit ring): parity, then A/B against gemm256 on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r39
timeout 300 python -m pytest tests/test_gpu_ops.py -k "gemm" -q --tb=short -p no:cacheprovider > gpurun_out/r39/pytest_gemm_default.log 2>&1; tail -4 gpurun_out/r39/pytest_gemm_default.log
WAN_GEMM_KERNEL=v4f timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q --tb=line -p no:cacheprovider > gpurun_out/r39/pytest_v4f.log 2>&1; tail -4 gpurun_out/r39/pytest_v4f.log
for k in v4 v3 v4; do
echo "== bench_gemm WAN_GEMM_KERNEL=$k"
WAN_GEMM_KERNEL=$k timeout 300 python tools/bench_gemm.py --rounds 4 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/r39/bench_gemm.log
