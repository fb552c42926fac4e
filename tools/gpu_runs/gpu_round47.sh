#!/bin/bash
# round-end validation: full GPU parity suite, smoke(), judged bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r47
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r47/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r47/pytest_gpu.log
tail -4 gpurun_out/r47/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r47/smoke.log
timeout 600 python bench.py > gpurun_out/r47/bench_14B.json 2> gpurun_out/r47/bench_14B.err; tail -1 gpurun_out/r47/bench_14B.json | cut -c1-400
