# the two experiment builds written after round 3's GPU budget was spent (neither had run on hardware): build them HERE first --
#   make -C wan2gp_amd/csrc persistent16 convwide
# -- then, in ONE call: the GEMM suites through libwanhip_mp.so (gemm256mp.hip), the VAE suites through libwanhip_conv.so
# (vae_conv256.inc) + the decode hash against the shipped kernel's, and the two A/B timings
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 500 python tools/pytest_with_lib.py libwanhip_mp.so tests/test_gpu_ops.py tests/test_gpu_baseline_configs.py tests/test_gpu_model.py -q -m gpu -x -p no:cacheprovider -k "gemm or block or forward" ) > $O/${ROUND}_pytest_gemm256mp_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_gemm256mp_$TAG.log
( timeout 500 python tools/pytest_with_lib.py libwanhip_conv.so tests/test_gpu_vae.py tests/test_gpu_vae_720p.py tests/test_gpu_vae22.py -q -m gpu -x -p no:cacheprovider ) > $O/${ROUND}_pytest_conv_wide_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_conv_wide_$TAG.log
for lib in libwanhip.so libwanhip_conv.so; do timeout 200 python tools/vae_hash.py --lib $lib 2>&1 | tail -2 | tee $O/${ROUND}_vae_hash_${lib%.so}_$TAG.log; done
for lib in libwanhip.so libwanhip_conv.so libwanhip.so libwanhip_conv.so; do timeout 200 python tools/bench_vae.py --lib $lib 2>&1 | tail -3 | tee -a $O/${ROUND}_bench_vae_ab_$TAG.log; done
for pass in 1 2; do for lib in libwanhip.so libwanhip_mp.so; do
timeout 300 python tools/gemm_vs_vendor.py --lib $lib --rounds 5 2>&1 | tee $O/${ROUND}_gemm_vs_vendor_${lib%.so}_pass${pass}_$TAG.log | grep "ours_over_vendor_median\|library"
done; done
# the fused epilogues (GELU, gated residual, V^T) are where a persistent tile walk could hide work: the same alternation on tools/bench_gemm.py
for pass in 1 2; do for lib in libwanhip.so libwanhip_mp.so; do
echo "== $lib pass $pass"; timeout 300 python tools/bench_gemm.py --lib $lib --rounds 5 2>&1 | tee $O/${ROUND}_bench_gemm_${lib%.so}_pass${pass}_$TAG.log | grep -o '"[a-z0-9+T]*": {"ms": [0-9.]*, "TF": [0-9.]*' 
done; done
