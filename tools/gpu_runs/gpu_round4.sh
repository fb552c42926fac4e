#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in pp pp_prio; do
  echo "== pytest attention variant $v"
  WAN_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -15
done
echo "== attn microbench self (L=75600,H=8,B=2)"
timeout 900 python tools/bench_attn.py --variants lean,lean8,pp,pp_prio --rounds 5 2>&1 | tee gpurun_out/bench_attn_self2.json | grep -E "TF_med|maxdiff|\"(lean|lean8|pp|pp_prio)\""
echo "== attn microbench 1.3B shape (L=32760,H=12,B=2)"
timeout 900 python tools/bench_attn.py --variants lean,lean8,pp,pp_prio --rounds 5 --L 32760 --H 12 2>&1 | tee gpurun_out/bench_attn_13b.json | grep -E "TF_med|maxdiff|\"(lean|lean8|pp|pp_prio)\""
echo "== attn microbench cross (Lk=512, H=40)"
timeout 900 python tools/bench_attn.py --Lk 512 --H 40 --rounds 6 --variants lean,lean8,pp,pp_prio 2>&1 | tee gpurun_out/bench_attn_cross2.json | grep -E "TF_med|maxdiff|\"(lean|lean8|pp|pp_prio)\""
echo "== PMC pp"
( cd /tmp && WAN_ATTN_VARIANT=pp timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace -d $OLDPWD/gpurun_out/pmc_attn_pp -o a --output-format csv -- python $OLDPWD/tools/bench_attn.py --variants pp --rounds 2 --H 4 ) > gpurun_out/pmc_attn_pp.log 2>&1
python tools/rocprof_summarize.py gpurun_out/pmc_attn_pp gpurun_out/pmc_attn_pp_summary.json "attn pp SQ counters" | grep attn_
