#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in w64p; do
echo "== pytest attention variant $v"
WAN_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -6
done
echo "== attn microbench self"
timeout 900 python tools/bench_attn.py --variants v2r_8,w64,w64p --rounds 5 2>&1 | tee gpurun_out/bench_attn_self9.json | grep -E "TF_med|maxdiff|\"(v2r_8|w64|w64p)\""
echo "== attn microbench cross"
timeout 900 python tools/bench_attn.py --Lk 512 --H 40 --rounds 6 --variants v2_4,w64,w64p 2>&1 | grep -E "TF_med|\"(v2_4|w64|w64p)\""
