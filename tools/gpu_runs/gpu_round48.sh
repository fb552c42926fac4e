#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r48
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_skipcache.py tests/test_gpu_sp.py tests/test_gpu_e2e.py tests/test_gpu_loader.py tests/test_gpu_t5.py -q -s --tb=short -p no:cacheprovider > gpurun_out/r48/pytest_vace.log 2>&1; echo "rc=$?" >> gpurun_out/r48/pytest_vace.log
grep -v "^$" gpurun_out/r48/pytest_vace.log | grep "vace\|passed\|failed\|Error\|rc=\|assert\|FAILED" | tail -20
