#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for g in 4 3 5 6; do
echo "== bench_gemm WAN_GEMM_GROUP=$g"
WAN_GEMM_GROUP=$g timeout 600 python tools/bench_gemm.py --rounds 4 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' '; echo
done
