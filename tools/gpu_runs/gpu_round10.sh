#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in v2_4 v2_8 v3_8 pp; do
echo "== pytest attention variant $v"
WAN_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -4
done
echo "== attn microbench self"
timeout 900 python tools/bench_attn.py --variants lean8,v2_4,v2_8,v3_8,pp --rounds 5 2>&1 | tee gpurun_out/bench_attn_self5.json | grep -E "TF_med|maxdiff|\"(lean8|v2_4|v2_8|v3_8|pp)\""
echo "== attn microbench cross"
timeout 900 python tools/bench_attn.py --Lk 512 --H 40 --rounds 6 --variants lean,v2_4,v2_8 2>&1 | tee gpurun_out/bench_attn_cross5.json | grep -E "TF_med|\"(lean|v2_4|v2_8)\""
