# round 6: the halo convolution's ring-stage race (a fragment read still pending when the stage's next weights land): one convolution launched
# 400 times in two processes side by side, the sharded tiled-VAE test repeated, output hashes and A/B timing against the previous build
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for lib in libwanhip_prev.so libwanhip.so; do echo "--- $lib:"; (timeout 300 python tools/probes/vae_conv_determinism.py A 400 --lib $lib & timeout 300 python tools/probes/vae_conv_determinism.py B 400 --lib $lib; wait) 2>&1 | grep "launches differ"; done | tee $O/${ROUND}_vae_conv_determinism_$TAG.log
for i in 1 2 3 4 5 6; do echo "sharded tiled VAE, run $i: $(timeout 250 python -m pytest tests/test_gpu_sp.py -q -k tiled_vae -p no:cacheprovider 2>&1 | tail -1)"; done | tee $O/${ROUND}_tiled_vae_sharded_repeat_$TAG.log
( echo "prev:"; timeout 300 python tools/vae_hash.py --lib libwanhip_prev.so; echo "new:"; timeout 300 python tools/vae_hash.py ) 2>&1 | grep -v amdgpu | tee $O/${ROUND}_vae_hash_prev_vs_new_$TAG.log
for i in 1 2; do
  echo "prev: $(timeout 300 python tools/bench_vae.py --encode --lib libwanhip_prev.so 2>/dev/null | tail -1)"
  echo "new:  $(timeout 300 python tools/bench_vae.py --encode 2>/dev/null | tail -1)"
done | tee $O/${ROUND}_ab_vae_$TAG.log
( timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae22.py tests/test_gpu_vae_720p.py -q -x -p no:cacheprovider ) 2>&1 | tail -2 | tee $O/${ROUND}_pytest_vae_$TAG.log
