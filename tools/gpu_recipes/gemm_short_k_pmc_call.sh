# why the short-K (Wan 1.3B) bf16 GEMMs run at 1150-1240 TFLOP/s where the long-K ones reach 1450-1500: true clock and matrix-pipe
# utilisation of gemm256m_kernel at both models' projection shapes (rocprofv3 --pmc in a pass of its own, --kernel-trace only;
# clock = GRBM_GUI_ACTIVE / 8 / time, utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles)), plus the L2-miss traffic of the short-K shapes
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
cd /tmp
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_g13 -o a --output-format csv -- python $R/tools/bench_gemm.py --model 1.3B --rounds 6 > $R/$O/pmc_g13.log 2>&1
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_g14 -o a --output-format csv -- python $R/tools/bench_gemm.py --model 14B --rounds 2 > $R/$O/pmc_g14.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pmc_g13f -o a --output-format csv -- python $R/tools/bench_gemm.py --model 1.3B --rounds 3 > $R/$O/pmc_g13f.log 2>&1
cd $R
python tools/rocprof_summarize.py $O/pmc_g13 $O/${ROUND}_gemm256m_1.3B_shapes_pmc_sq_$TAG.json "tools/bench_gemm.py --model 1.3B (M = 65,520; d 1536, ffn 8960), SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/pmc_g14 $O/${ROUND}_gemm256m_14B_shapes_pmc_sq_$TAG.json "tools/bench_gemm.py --model 14B (M = 151,200; d 5120, ffn 13824), SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/pmc_g13f $O/${ROUND}_gemm256m_1.3B_shapes_pmc_FETCH_SIZE_$TAG.json "tools/bench_gemm.py --model 1.3B, FETCH_SIZE pass (x2 on gfx950)" > /dev/null
rm -rf $O/pmc_g13 $O/pmc_g14 $O/pmc_g13f
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_pmc_*.json")):
    j = json.load(open(f))
    for k, v in j["kernels"].items():
        if "gemm" not in k:
            continue
        c = {n: x["avg"] for n, x in j["counters"].get(k, {}).items()}
        g = c.get("GRBM_GUI_ACTIVE", 0) / 8
        if g and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0):
            print(f.split("/")[-1][:40], k[:60], v["calls"], round(v["avg_ms"], 3), "clock %.3f GHz" % (g / v["avg_ms"] / 1e6), "mfma busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g)))
        if "FETCH_SIZE" in c:
            print(f.split("/")[-1][:40], k[:60], v["calls"], round(v["avg_ms"], 3), "FETCH x2 %.1f MB" % (c["FETCH_SIZE"] * 2 / 1000))
PY
