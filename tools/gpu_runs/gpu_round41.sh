#!/bin/bash
# hardware bf16 rounding everywhere + gemm256k: full parity suite, GEMM shapes, tile timeline, 14B bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r41
timeout 120 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed on this box"; exit 0; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r41/pytest_gpu.log 2>&1; tail -5 gpurun_out/r41/pytest_gpu.log
echo "== bench_gemm v4"; timeout 300 python tools/bench_gemm.py --rounds 4 2>&1 | grep -E "\"|TF" | paste - - - | sed 's/  */ /g' | awk '{print $1, $NF}' | tr '\n' ' ' | tee gpurun_out/r41/bench_gemm.log; echo
timeout 200 python tools/gemm_stamp.py 2>&1 | grep timeline | tail -4 | tee gpurun_out/r41/stamps.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/r41/bench_14B.json 2> gpurun_out/r41/bench_14B.err; tail -1 gpurun_out/r41/bench_14B.json
