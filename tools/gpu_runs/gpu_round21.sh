#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest attention variant w64f"
WAN_ATTN_VARIANT=w64f timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "attention" 2>&1 | tail -6
echo "== attn microbench self"
timeout 600 python tools/bench_attn.py --variants w64,w64q,w64f --rounds 5 --stamps w64qt,w64ft 2>&1 | tee gpurun_out/bench_attn_self11.json | grep -E "stamps|TF_med|maxdiff|\"(w64|w64q|w64f)\""
echo "== attn microbench cross"
timeout 600 python tools/bench_attn.py --Lk 512 --H 40 --rounds 6 --variants v2_4,w64q,w64f 2>&1 | grep -E "TF_med|\"(v2_4|w64f|w64q)\""
