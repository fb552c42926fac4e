#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "norm or rope or residual or layernorm" 2>&1 | tail -3
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['other_kernels'])"
