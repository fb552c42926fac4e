# round 4: the VAE convolution, round-4 variants against the 128-pixel two-stage form (libwanhip_cn.so, make cn):
# K-step): bit identity against the previous build (libwanhip_cn.so = HEAD's vae_ops.hip, built by hand), the VAE suites, alternating timings
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
for lib in libwanhip_cn.so libwanhip.so; do timeout 300 python tools/vae_hash.py --lib $lib 2>&1 | tail -1 | tee $O/${ROUND}_vae_hash_${lib%.so}_$TAG.log; done
cmp $O/${ROUND}_vae_hash_libwanhip_cn_$TAG.log $O/${ROUND}_vae_hash_libwanhip_$TAG.log && echo "VAE OUTPUTS BIT-IDENTICAL" | tee $O/${ROUND}_vae_bit_identity_$TAG.log
( timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae_720p.py tests/test_gpu_vae22.py -q -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_vae_conv_$TAG.log 2>&1; tail -3 $O/${ROUND}_pytest_vae_conv_$TAG.log
for pass in 1 2; do for lib in libwanhip_cn.so libwanhip.so; do echo "== $lib"; timeout 200 python tools/bench_vae.py --lib $lib --encode 2>&1 | tail -1 | tee -a $O/${ROUND}_bench_vae_ab_$TAG.log; done; done
