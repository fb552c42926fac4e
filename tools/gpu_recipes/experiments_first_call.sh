# the two experiment builds written after round 3's GPU budget was spent (neither has run on hardware): build them HERE first --
#   make -C wan2gp_amd/csrc persistent16 convwide
# -- then, in ONE call: the GEMM suites through libwanhip_mp.so (gemm256mp.hip), the VAE suites through libwanhip_conv.so
# (vae_conv256.inc) + the decode hash against the shipped kernel's, and the two A/B timings
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
# the three -m gpu tests written behind round 3's last GPU run (CFG parallelism on one GPU, the LoRA extract -> merge round trip): shipped library
( timeout 600 python -m pytest tests/test_gpu_zzz_cfg_parallel.py tests/test_gpu_zzz_lora_extract.py -q -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_late_tests_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_late_tests_$TAG.log
( timeout 600 python tools/pytest_with_lib.py libwanhip_mp.so tests/test_gpu_ops.py tests/test_gpu_baseline_configs.py tests/test_gpu_model.py -q -m gpu -x -p no:cacheprovider -k "gemm or block or forward" ) > $O/${ROUND}_pytest_gemm256mp_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_gemm256mp_$TAG.log
( timeout 600 python tools/pytest_with_lib.py libwanhip_conv.so tests/test_gpu_vae.py tests/test_gpu_vae_720p.py tests/test_gpu_vae22.py -q -m gpu -x -p no:cacheprovider ) > $O/${ROUND}_pytest_conv_wide_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_conv_wide_$TAG.log
for lib in libwanhip.so libwanhip_conv.so; do timeout 200 python tools/vae_hash.py --lib $lib 2>&1 | tail -2 | tee $O/${ROUND}_vae_hash_${lib%.so}_$TAG.log; done
for lib in libwanhip.so libwanhip_conv.so libwanhip.so libwanhip_conv.so; do timeout 200 python tools/bench_vae.py --lib $lib 2>&1 | tail -3 | tee -a $O/${ROUND}_bench_vae_ab_$TAG.log; done
for pass in 1 2; do for lib in libwanhip.so libwanhip_mp.so; do
timeout 300 python tools/gemm_vs_vendor.py --lib $lib --rounds 5 2>&1 | tee $O/${ROUND}_gemm_vs_vendor_${lib%.so}_pass${pass}_$TAG.log | grep "ours_over_vendor_median\|library"
done; done
