#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r46
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_skipcache.py tests/test_gpu_e2e.py -q -s --tb=short -p no:cacheprovider > gpurun_out/r46/pytest_i2v21.log 2>&1; echo "rc=$?" >> gpurun_out/r46/pytest_i2v21.log
grep -v "^$" gpurun_out/r46/pytest_i2v21.log | grep "i2v21\|passed\|failed\|Error\|rc=\|assert" | tail -20
