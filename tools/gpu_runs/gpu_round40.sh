#!/bin/bash
# gemm256k with the LDS-transposed epilogue: parity, A/B, tile timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r40
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item(), torch.cuda.get_device_name(0))" || { echo "GPU init failed on this box"; exit 0; }
WAN_GEMM_KERNEL=v4f timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q --tb=short -p no:cacheprovider > gpurun_out/r40/pytest_v4f.log 2>&1; tail -4 gpurun_out/r40/pytest_v4f.log
timeout 300 python -m pytest tests/test_gpu_ops.py -k "gemm" -q --tb=short -p no:cacheprovider > gpurun_out/r40/pytest_gemm_default.log 2>&1; tail -2 gpurun_out/r40/pytest_gemm_default.log
for k in v4 v3 v4; do
echo "== bench_gemm WAN_GEMM_KERNEL=$k"
WAN_GEMM_KERNEL=$k timeout 300 python tools/bench_gemm.py --rounds 4 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/r40/bench_gemm.log
timeout 200 python tools/gemm_stamp.py 2>&1 | grep timeline | tee gpurun_out/r40/stamps.log
