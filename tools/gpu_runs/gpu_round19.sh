#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest attention variant w64q"
WAN_ATTN_VARIANT=w64q timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -25
echo "== attn microbench self"
timeout 600 python tools/bench_attn.py --variants v2r_8,w64,w64q --rounds 5 --stamps w64t,w64qt 2>&1 | tee gpurun_out/bench_attn_self10.json | grep -E "stamps|TF_med|maxdiff|\"(v2r_8|w64|w64q)\""
echo "== attn microbench cross"
timeout 600 python tools/bench_attn.py --Lk 512 --H 40 --rounds 6 --variants v2_4,w64,w64q 2>&1 | grep -E "TF_med|\"(v2_4|w64|w64q)\""
