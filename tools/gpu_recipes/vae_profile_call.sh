# where a 720p x 81-frame VAE decode spends its time (kernel trace), and what the convolution kernel is waiting for (SQ counters, own pass)
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/vae_trace -o a --output-format csv -- python $R/tools/bench_vae.py > $R/$O/vae_trace.log 2>&1
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $PMC --kernel-trace -d $R/$O/vae_pmc -o a --output-format csv -- python $R/tools/bench_vae.py > $R/$O/vae_pmc.log 2>&1
PMC2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $PMC2 --kernel-trace -d $R/$O/vae_pmc2 -o a --output-format csv -- python $R/tools/bench_vae.py > $R/$O/vae_pmc2.log 2>&1
cd $R
python tools/rocprof_summarize.py $O/vae_trace $O/${ROUND}_vae_decode_720p_kernel_trace_$TAG.json "tools/bench_vae.py: two 720p x 81f decodes" > /dev/null
python tools/rocprof_summarize.py $O/vae_pmc $O/${ROUND}_vae_conv_pmc_sq_$TAG.json "tools/bench_vae.py, SQ / GRBM pass" > /dev/null
python tools/rocprof_summarize.py $O/vae_pmc2 $O/${ROUND}_vae_conv_pmc_lds_$TAG.json "tools/bench_vae.py, LDS / issue pass" > /dev/null
rm -rf $O/vae_trace $O/vae_pmc $O/vae_pmc2
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*vae*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1])
    for k, v in list(j["kernels"].items())[:8]:
        c = {n: round(x["avg"], 1) for n, x in j["counters"].get(k, {}).items()}
        print("  ", k[:70], v["calls"], v["total_ms"], v["pct"], c)
PY
