# round 6: the rotation on aligned register pairs (no packed-f32 instruction reads a pair crosswise any more) + plain adds in the LayerNorm / gate
# modulation sums: row-kernel, mixed-plan and forward tests, the same bits as the previous build (tools/rows_hash.py), A/B timing, and the
# LDS-less narrow kernel with the new rotation (rrwg0f1) beside a neighbour that comes and goes (rrwg0 = rounds 3-5: the control)
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( echo "prev:"; timeout 200 python tools/rows_hash.py --lib libwanhip_prev.so; echo "new:"; timeout 200 python tools/rows_hash.py ) 2>&1 | grep -v amdgpu | tee $O/${ROUND}_rows_hash_prev_vs_new_$TAG.log
for i in 1 2; do echo "prev: $(timeout 200 python tools/bench_rows.py --lib libwanhip_prev.so 2>/dev/null | tr -d "\n " | cut -c1-600)"; echo "new:  $(timeout 200 python tools/bench_rows.py 2>/dev/null | tr -d "\n " | cut -c1-600)"; done | tee $O/${ROUND}_ab_rows_$TAG.log
for lib in libwanhip_rrwg0f1.so libwanhip_rrwg0.so; do
  echo "--- $lib"
  ( for n in 1 2 3 4; do timeout 60 python tools/probes/dit_determinism.py CO 6 big > /dev/null 2>&1; done ) &
  NB=$!
  timeout 150 python tools/probes/rmsrope_twice.py V 50 --lib $lib 2>&1 | grep -E "launches differ|rror" | cut -c1-300
  wait $NB
done | tee $O/${ROUND}_rmsrope_aligned_rotation_$TAG.log
( timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_mixed.py tests/test_gpu_model.py tests/test_gpu_nag.py -q -x -p no:cacheprovider ) 2>&1 | tail -3 | tee $O/${ROUND}_pytest_ops_mixed_model_nag_$TAG.log
