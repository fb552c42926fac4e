#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r45
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
timeout 600 python -m pytest tests/test_gpu_skipcache.py tests/test_gpu_model.py tests/test_gpu_sp.py -q -s --tb=short -p no:cacheprovider > gpurun_out/r45/pytest_skipcache.log 2>&1; echo "rc=$?" >> gpurun_out/r45/pytest_skipcache.log
grep -v "^$" gpurun_out/r45/pytest_skipcache.log | grep -v "err_ref\|hip-vs-ref" | tail -30
