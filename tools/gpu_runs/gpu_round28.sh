#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for d in 1 2; do
echo "== gemm32 DIST=$d: tests + bench"
WAN_GEMM32_DIST=$d timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "gemm or linear" 2>&1 | tail -2
WAN_GEMM32_DIST=$d timeout 600 python tools/bench_gemm.py --rounds 5 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g'
done
