# round 6: which half of the cure cures?  The narrow in-place RMSNorm + RoPE kernel with neither the LDS word nor the barrier (rrwg0 = rounds 3-5),
# the LDS word alone (rrwg1), the barrier alone (rrwg2) and the shipped form (both), each launched for 60 s on fresh copies of fixed rows while
# neighbour processes (a 2-layer forward at the 1.3B widths) start, run and exit one after the other.  `make -C wan2gp_amd/csrc rrwg` first.
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
for lib in libwanhip_rrwg0.so libwanhip_rrwg1.so libwanhip_rrwg2.so libwanhip.so; do
  echo "--- $lib"
  ( for n in 1 2 3 4; do timeout 60 python tools/probes/dit_determinism.py CO 6 big > /dev/null 2>&1; done ) &
  NB=$!
  timeout 150 python tools/probes/rmsrope_twice.py V 60 --lib $lib 2>&1 | grep -E "launches differ|iteration" | cut -c1-260
  wait $NB
done | tee $O/${ROUND}_rmsrope_cure_in_parts_$TAG.log
