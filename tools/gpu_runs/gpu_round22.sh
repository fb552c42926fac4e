#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in w64 w64q w64f; do
echo "== pytest attention variant $v"
WAN_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "attention" 2>&1 | tail -4
done
echo "== attn microbench self"
timeout 600 python tools/bench_attn.py --variants w64,w64q,w64f --rounds 5 --stamps w64qt,w64ft 2>&1 | tee gpurun_out/bench_attn_self12.json | grep -E "stamps|TF_med|maxdiff|\"(w64|w64q|w64f)\""
