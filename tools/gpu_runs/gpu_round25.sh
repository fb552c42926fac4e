#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for t in 256 128; do
echo "== pytest gemm, WAN_GEMM_TILE=$t"
WAN_GEMM_TILE=$t timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vae.py -m gpu -q --tb=line -p no:cacheprovider -k "gemm or linear or vae" 2>&1 | tail -4
echo "== bench_gemm WAN_GEMM_TILE=$t"
WAN_GEMM_TILE=$t timeout 600 python tools/bench_gemm.py --rounds 5 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g'
done
