#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest vae"
timeout 900 python -m pytest tests/test_gpu_vae.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | tee gpurun_out/pytest_vae.log | tail -40
