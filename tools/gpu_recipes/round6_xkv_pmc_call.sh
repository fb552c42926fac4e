# counters on the text cross-attention kernel (round 6: attention_xkv.hip) at the bench shape (B 2, H 40, Lq 75,600, 512 keys): true clock =
# GRBM_GUI_ACTIVE / 8 / time, matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles) -- is the 0.34 a stalled pipe at a high clock
# (room for software) or a busy one at a low clock (the power limit, like the GEMMs)?  --pmc in a pass of its own, --kernel-trace only
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
cd /tmp
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_x -o a --output-format csv -- python $R/tools/bench_attn.py --Lk 512 --modes bounded --rounds 8 > $R/$O/pmc_x.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $R/$O/pmc_x2 -o a --output-format csv -- python $R/tools/bench_attn.py --Lk 512 --modes bounded --rounds 8 > $R/$O/pmc_x2.log 2>&1
cd $R
python tools/rocprof_summarize.py $O/pmc_x $O/${ROUND}_cross_attention_xkv_pmc_sq_$TAG.json "tools/bench_attn.py --Lk 512 --modes bounded (attention_xkv.hip), SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/pmc_x2 $O/${ROUND}_cross_attention_xkv_pmc_insts_$TAG.json "the same, instruction-class pass" > /dev/null
rm -rf $O/pmc_x $O/pmc_x2
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*cross_attention_xkv_pmc*.json")):
    j = json.load(open(f))
    for k, v in j["kernels"].items():
        if "attn" not in k:
            continue
        c = {n: x["avg"] for n, x in j["counters"].get(k, {}).items()}
        g = c.get("GRBM_GUI_ACTIVE", 0) / 8
        print(f.split("/")[-1][:50], k[:50], v["calls"], round(v["avg_ms"], 3), ("clock %.3f GHz mfma busy %.3f" % (g / v["avg_ms"] / 1e6, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g))) if g else "", {n: round(x) for n, x in c.items()})
PY
