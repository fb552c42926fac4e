#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6 | tee gpurun_out/r24_pytest_gpu.log
echo "== bench 14B flat"
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 | tee gpurun_out/r24_bench_14B.json | cut -c1-1100
echo "== bench 14B gemm v1 (A/B)"
WAN_GEMM_KERNEL=v1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r24_bench_14B_gemmv1.json | cut -c1-400
