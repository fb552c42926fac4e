#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15 | tee gpurun_out/r20_pytest_gpu.log
echo "== attention tests forced through w64 / w64q"
for v in w64 w64q; do WAN_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "attention" 2>&1 | tail -4; done
echo "== bench 14B"
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -3 | tee gpurun_out/r20_bench_14B.json
echo "== bench 14B exact-qscale (A/B)"
WAN_DIT_EXACT_QSCALE=1 timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 | tee gpurun_out/r20_bench_14B_exact.json
