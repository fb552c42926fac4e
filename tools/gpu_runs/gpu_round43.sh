#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r43
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
timeout 600 python -m pytest tests/test_gpu_vae22.py tests/test_gpu_vae.py -q -s --tb=short -p no:cacheprovider > gpurun_out/r43/pytest_vae22.log 2>&1; echo "rc=$?" >> gpurun_out/r43/pytest_vae22.log
grep -v "^$" gpurun_out/r43/pytest_vae22.log | tail -30
