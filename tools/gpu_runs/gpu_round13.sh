#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in v2_8 v4_8 v2r_8; do
echo "== PMC $v"
( cd /tmp && WAN_ATTN_VARIANT=$v timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d $OLDPWD/gpurun_out/pmc_$v -o a --output-format csv -- python $OLDPWD/tools/bench_attn.py --variants $v --rounds 2 --H 4 ) > gpurun_out/pmc_$v.log 2>&1
python tools/rocprof_summarize.py gpurun_out/pmc_$v gpurun_out/pmc_${v}_summary.json "attn $v" | grep -E "attn_pp|avg_ms" | head -3
python - <<PY
import json
d=json.load(open("gpurun_out/pmc_${v}_summary.json"))
for k,v in d["kernels"].items():
    if "attn" in k: print(k, v)
PY
done
