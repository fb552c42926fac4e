#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest attention variant v3_8"
WAN_ATTN_VARIANT=v3_8 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -8
echo "== attn microbench self"
timeout 900 python tools/bench_attn.py --variants lean8,v2_8,v3_8 --rounds 5 2>&1 | tee gpurun_out/bench_attn_self4.json | grep -E "TF_med|maxdiff|\"(lean8|v2_8|v3_8)\""
echo "== gemm microbench"
timeout 900 python tools/bench_gemm.py 2>&1 | tee gpurun_out/bench_gemm.json | tail -20
echo "== PMC traffic, 14B self-attention shape (B=2,L=75600,H=40), default kernel"
for c in FETCH_SIZE WRITE_SIZE; do
 ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace -d $OLDPWD/gpurun_out/pmc14_$c -o a --output-format csv -- python $OLDPWD/tools/bench_attn.py --variants v2_8 --rounds 1 --H 40 ) > gpurun_out/pmc14_$c.log 2>&1
 python tools/rocprof_summarize.py gpurun_out/pmc14_$c gpurun_out/r01_14B_attn_pmc_$c.json "14B attn $c" | grep attn_pp
done
