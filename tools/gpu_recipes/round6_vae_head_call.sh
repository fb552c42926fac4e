# round 6: the decoder head (96 -> 3 channels) on a 16-channel tile of the halo convolution -- output hashes against the previous build, the VAE
# suites, A/B timing on one box (previous / new / previous / new)
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( echo "prev:"; timeout 300 python tools/vae_hash.py --lib libwanhip_prev.so; echo "new:"; timeout 300 python tools/vae_hash.py ) > $O/${ROUND}_vae_hash_prev_vs_head16_$TAG.log 2>&1; cat $O/${ROUND}_vae_hash_prev_vs_head16_$TAG.log | grep -v amdgpu
( timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae22.py tests/test_gpu_vae_720p.py -q -x -p no:cacheprovider ) > $O/${ROUND}_pytest_vae_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_vae_$TAG.log
for i in 1 2; do
  echo "prev:" >> $O/${ROUND}_ab_vae_head16_$TAG.log; timeout 300 python tools/bench_vae.py --encode --lib libwanhip_prev.so 2>/dev/null | tail -1 >> $O/${ROUND}_ab_vae_head16_$TAG.log
  echo "new:" >> $O/${ROUND}_ab_vae_head16_$TAG.log; timeout 300 python tools/bench_vae.py --encode 2>/dev/null | tail -1 >> $O/${ROUND}_ab_vae_head16_$TAG.log
done
cat $O/${ROUND}_ab_vae_head16_$TAG.log
