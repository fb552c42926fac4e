#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== gemm tests forced through gemm256 (v3f)"
WAN_GEMM_KERNEL=v3f timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -3
echo "== bench_gemm default"
timeout 600 python tools/bench_gemm.py --rounds 5 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' '; echo
echo "== bench_gemm v2 (same box reference)"
WAN_GEMM_KERNEL=v2 timeout 600 python tools/bench_gemm.py --rounds 5 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' '; echo
