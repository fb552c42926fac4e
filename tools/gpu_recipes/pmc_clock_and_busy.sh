# counters (own pass: --pmc with --kernel-trace only) on the shipped attention loop, the fp8 GEMM and the bf16 GEMM beside the vendor kernel
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
cd /tmp
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_attn -o a --output-format csv -- python $R/tools/bench_attn.py --rounds 1 --modes bounded > $R/$O/pmc_attn.log 2>&1
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_fp8 -o a --output-format csv -- python $R/tools/bench_fp8.py > $R/$O/pmc_fp8.log 2>&1
for s in ffn1 qkvo; do
timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/pmc_$s -o a --output-format csv -- python $R/tools/gemm_pmc_pair.py --shape $s > $R/$O/pmc_$s.log 2>&1
done
cd $R
python tools/rocprof_summarize.py $O/pmc_attn $O/${ROUND}_14B_attn_pmc_sq_$TAG.json "shipped bounded self-attention at B=2 H=40 L=75600 (tools/bench_attn.py --rounds 1 --modes bounded), SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/pmc_fp8 $O/${ROUND}_gemm_fp8_pmc_sq_$TAG.json "tools/bench_fp8.py (M=151200: qkv, ffn1+GELU, ffn2), SQ / GRBM pass" > /dev/null
for s in ffn1 qkvo; do
python tools/rocprof_summarize.py $O/pmc_$s $O/${ROUND}_gemm_vendor_vs_ours_pmc_${s}_$TAG.json "$s at M=151200: torch.addmm (hipBLASLt) and the shipped GEMM alternating on the same tensors (tools/gemm_pmc_pair.py), SQ / GRBM pass" > /dev/null
done
rm -rf $O/pmc_attn $O/pmc_fp8 $O/pmc_ffn1 $O/pmc_qkvo
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_pmc_*.json")):
    j = json.load(open(f))
    for k, v in j["kernels"].items():
        c = {n: x["avg"] for n, x in j["counters"].get(k, {}).items()}
        g = c.get("GRBM_GUI_ACTIVE", 0) / 8
        if g and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0):
            print(f.split("/")[-1][:40], k[:44], v["calls"], v["avg_ms"], "clock %.3f GHz" % (g / v["avg_ms"] / 1e6),
                  "mfma busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g)))
PY
