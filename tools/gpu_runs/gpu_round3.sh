#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest attention (default variant)"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -5
for v in lean_prio lean8 lean8_prio; do
  echo "== pytest attention variant $v"
  WAN_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "attention" 2>&1 | tail -3
done
echo "== attn microbench self (L=75600,H=8,B=2)"
timeout 900 python tools/bench_attn.py 2>&1 | tee gpurun_out/bench_attn_self.json | tail -40
echo "== attn microbench cross (Lk=512, H=40)"
timeout 900 python tools/bench_attn.py --Lk 512 --H 40 --rounds 6 2>&1 | tee gpurun_out/bench_attn_cross.json | tail -40
echo "== counters list"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|GRBM)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' | head -c 6000 > gpurun_out/counters_list.txt; wc -c gpurun_out/counters_list.txt
echo "== PMC on attention microbench (lean)"
( cd /tmp && WAN_ATTN_VARIANT=lean timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $OLDPWD/gpurun_out/pmc_attn -o a --output-format csv -- python $OLDPWD/tools/bench_attn.py --variants lean --rounds 2 --H 4 ) > gpurun_out/pmc_attn.log 2>&1
tail -2 gpurun_out/pmc_attn.log
python tools/rocprof_summarize.py gpurun_out/pmc_attn gpurun_out/pmc_attn_summary.json "attn lean SQ counters" | tail -8
