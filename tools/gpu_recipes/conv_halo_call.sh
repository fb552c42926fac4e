# round 4: the halo-patch convolution (csrc/vae_conv_halo.hip) -- parity (fp64 + the gather kernel), decode / encode timings alternating, the VAE suites,
# and (PMC=1) the LDS / matrix-pipe counters of a decode
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_vae_720p.py -q -m gpu -p no:cacheprovider -k "halo or big_offsets or conv2d_3x3" -x ) > $O/${ROUND}_pytest_conv_halo_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_conv_halo_$TAG.log
for pass in 1 2; do for m in "--no-halo" ""; do echo "== halo ${m:-on}" | tee -a $O/${ROUND}_bench_vae_halo_ab_$TAG.log; timeout 200 python tools/bench_vae.py --encode $m 2>&1 | tail -1 | tee -a $O/${ROUND}_bench_vae_halo_ab_$TAG.log; done; done
if [ -n "$PMC" ]; then
cd /tmp
PMC1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $PMC1 --kernel-trace -d $R/$O/vae_pmc -o a --output-format csv -- python $R/tools/bench_vae.py > $R/$O/vae_pmc.log 2>&1
PMC2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $PMC2 --kernel-trace -d $R/$O/vae_pmc2 -o a --output-format csv -- python $R/tools/bench_vae.py > $R/$O/vae_pmc2.log 2>&1
cd $R
python tools/rocprof_summarize.py $O/vae_pmc $O/${ROUND}_vae_conv_pmc_sq_$TAG.json "tools/bench_vae.py, SQ / GRBM pass" | head -30
python tools/rocprof_summarize.py $O/vae_pmc2 $O/${ROUND}_vae_conv_pmc_lds_$TAG.json "tools/bench_vae.py, LDS / issue pass" | tail -12
rm -rf $O/vae_pmc $O/vae_pmc2
fi
if [ -n "$SUITES" ]; then ( timeout 1200 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae_720p.py tests/test_gpu_vae22.py -q -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_vae_halo_$TAG.log 2>&1; tail -8 $O/${ROUND}_pytest_vae_halo_$TAG.log; fi
true
