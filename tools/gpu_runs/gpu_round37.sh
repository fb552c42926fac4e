#!/bin/bash
# loader / LoRA row parity
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r37
timeout 600 python -m pytest tests/test_gpu_loader.py -q > gpurun_out/r37/pytest_loader.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r37/pytest_loader.log
tail -25 gpurun_out/r37/pytest_loader.log
