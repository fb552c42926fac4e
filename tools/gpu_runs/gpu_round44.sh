#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r44
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r44/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r44/pytest_gpu.log
grep -v "^$" gpurun_out/r44/pytest_gpu.log | tail -25
