#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest gemm/linear/vae/model (gemm32 default)"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vae.py tests/test_gpu_model.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -6
echo "== bench_gemm gemm32"
timeout 600 python tools/bench_gemm.py --rounds 5 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g'
echo "== bench_gemm v1"
WAN_GEMM_KERNEL=v1 timeout 600 python tools/bench_gemm.py --rounds 5 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g'
