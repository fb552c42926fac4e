# round 4 counters (own passes: --pmc with --kernel-trace only): the plain and the shifted self-attention loop, the fp8 GEMM of
# gemm_fp8m.hip (SQ / GRBM: true clock = GRBM_GUI_ACTIVE / 8 / time, matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles)),
# and RMSNorm+RoPE's HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes, MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950)
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
cd /tmp
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_attn1 -o a --output-format csv -- python $R/tools/bench_attn.py --rounds 1 --modes bounded > $R/$O/pmc_attn1.log 2>&1
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_attn12 -o a --output-format csv -- python $R/tools/bench_attn.py --rounds 1 --modes bounded --gain 12 > $R/$O/pmc_attn12.log 2>&1
timeout 300 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_fp8 -o a --output-format csv -- python $R/tools/bench_fp8.py > $R/$O/pmc_fp8.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pmc_rows_f -o a --output-format csv -- python $R/tools/bench_rows.py --rounds 3 > $R/$O/pmc_rows_f.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pmc_rows_w -o a --output-format csv -- python $R/tools/bench_rows.py --rounds 3 > $R/$O/pmc_rows_w.log 2>&1
cd $R
python tools/rocprof_summarize.py $O/pmc_attn1 $O/${ROUND}_14B_attn_plain_pmc_sq_$TAG.json "plain bounded self-attention at B=2 H=40 L=75600, K gain 1 (tools/bench_attn.py --rounds 1 --modes bounded), SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/pmc_attn12 $O/${ROUND}_14B_attn_shifted_pmc_sq_$TAG.json "the same launch at K gain 12: every workgroup in the shifted twin (FLAGS | 128), SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/pmc_fp8 $O/${ROUND}_gemm_fp8m_pmc_sq_$TAG.json "tools/bench_fp8.py (M=151200: qkv, ffn1+GELU, ffn2, o+gate) on gemm_fp8m.hip, SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/pmc_rows_f $O/${ROUND}_row_kernels_pmc_FETCH_SIZE_$TAG.json "tools/bench_rows.py: RMSNorm+RoPE and LN+modulate at 14B-720p / 1.3B-480p, FETCH_SIZE pass (x2 on gfx950)" > /dev/null
python tools/rocprof_summarize.py $O/pmc_rows_w $O/${ROUND}_row_kernels_pmc_WRITE_SIZE_$TAG.json "the same, WRITE_SIZE pass" > /dev/null
rm -rf $O/pmc_attn1 $O/pmc_attn12 $O/pmc_fp8 $O/pmc_rows_f $O/pmc_rows_w
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_pmc_*.json")):
    j = json.load(open(f))
    for k, v in j["kernels"].items():
        c = {n: x["avg"] for n, x in j["counters"].get(k, {}).items()}
        g = c.get("GRBM_GUI_ACTIVE", 0) / 8
        if g and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) and v["avg_ms"] > 1:
            print(f.split("/")[-1][:44], k[:44], v["calls"], v["avg_ms"], "clock %.3f GHz" % (g / v["avg_ms"] / 1e6),
                  "mfma busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g)))
        for n in ("FETCH_SIZE", "WRITE_SIZE"):
            if n in c and v["avg_ms"] > 0.2:
                print(f.split("/")[-1][:44], k[:44], v["calls"], v["avg_ms"], n, "%.1f MB (KB counter x 1024%s)" % (c[n] * 1024 * (2 if n == "FETCH_SIZE" else 1) / 1e6, ", x2" if n == "FETCH_SIZE" else ""))
PY
