#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== gemm tests forced through gemm256 (v3f)"
WAN_GEMM_KERNEL=v3f timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "gemm or linear" 2>&1 | tail -3
for k in v3 v3f; do
echo "== bench_gemm WAN_GEMM_KERNEL=$k"
WAN_GEMM_KERNEL=$k timeout 600 python tools/bench_gemm.py --rounds 5 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g'
done
