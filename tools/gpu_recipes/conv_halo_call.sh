# round 4: the halo-patch convolution (csrc/vae_conv_halo.hip) -- parity (fp64 + the gather kernel), the VAE suites, decode / encode timings alternating
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_vae_720p.py -q -m gpu -p no:cacheprovider -k "halo or big_offsets" -x ) > $O/${ROUND}_pytest_conv_halo_$TAG.log 2>&1; tail -15 $O/${ROUND}_pytest_conv_halo_$TAG.log
for pass in 1 2; do for m in "--no-halo" ""; do echo "== halo ${m:-on}" | tee -a $O/${ROUND}_bench_vae_halo_ab_$TAG.log; timeout 200 python tools/bench_vae.py --encode $m 2>&1 | tail -1 | tee -a $O/${ROUND}_bench_vae_halo_ab_$TAG.log; done; done
( timeout 1200 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae_720p.py tests/test_gpu_vae22.py -q -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_vae_halo_$TAG.log 2>&1; tail -8 $O/${ROUND}_pytest_vae_halo_$TAG.log
