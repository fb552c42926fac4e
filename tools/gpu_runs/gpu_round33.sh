#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for k in v1 v2 v3f; do echo "== 1.3B shapes, WAN_GEMM_KERNEL=$k"; WAN_GEMM_KERNEL=$k timeout 300 python tools/bench_gemm_small.py 2>&1 | tail -1; done
