#!/bin/bash
# T5 row: parity tests + full-geometry timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r36
timeout 600 python -m pytest tests/test_gpu_t5.py -q -s > gpurun_out/r36/pytest_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r36/pytest_t5.log
tail -15 gpurun_out/r36/pytest_t5.log
