#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in v2_4 v2_8; do
  echo "== pytest attention variant $v"
  WAN_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -8
done
echo "== attn microbench self (L=75600,H=8,B=2)"
timeout 900 python tools/bench_attn.py --variants lean,lean8,v2_4,v2_8 --rounds 5 2>&1 | tee gpurun_out/bench_attn_self3.json | grep -E "TF_med|maxdiff|\"(lean|lean8|v2_4|v2_8)\""
echo "== attn microbench cross (Lk=512, H=40)"
timeout 900 python tools/bench_attn.py --Lk 512 --H 40 --rounds 6 --variants lean,lean8,v2_4,v2_8 2>&1 | tee gpurun_out/bench_attn_cross3.json | grep -E "TF_med|maxdiff|\"(lean|lean8|v2_4|v2_8)\""
echo "== full pytest -m gpu (default variants) incl SP"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -25
