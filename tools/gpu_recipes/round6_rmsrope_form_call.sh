# round 6: the LDS-less narrow in-place RMSNorm + RoPE kernel (rrwg0) with the rotation's crosswise products (the ones found missing: run 82) computed
# otherwise -- two plain v_mul_f32 (rrwg0f1), the packed multiply never onto its own source pair (rrwg0f2) -- beside a neighbour that comes and goes
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
for lib in libwanhip_rrwg0f1.so libwanhip_rrwg0f2.so libwanhip_rrwg0.so; do
  echo "--- $lib"
  ( for n in 1 2 3 4; do timeout 60 python tools/probes/dit_determinism.py CO 6 big > /dev/null 2>&1; done ) &
  NB=$!
  timeout 150 python tools/probes/rmsrope_twice.py V 60 --lib $lib 2>&1 | grep -E "launches differ|wrong chunks equal|rror" | cut -c1-420 | head -5
  wait $NB
done | tee $O/${ROUND}_rmsrope_rotation_forms_$TAG.log
