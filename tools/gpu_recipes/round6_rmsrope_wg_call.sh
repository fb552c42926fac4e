# round 6: the narrow in-place RMSNorm + RoPE kernel made a workgroup (one LDS word per wave + one barrier): the row-kernel tests, the same
# bits as the previous build (tools/rows_hash.py), A/B timing, and the determinism probes beside a neighbour (the whole forward at 1.3B widths, the kernel alone)
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "rmsnorm or rope or row" ) 2>&1 | tail -2 | tee $O/${ROUND}_pytest_rows_$TAG.log
( echo "prev:"; timeout 200 python tools/rows_hash.py --lib libwanhip_prev.so; echo "new:"; timeout 200 python tools/rows_hash.py ) 2>&1 | grep -v amdgpu | tee $O/${ROUND}_rows_hash_prev_vs_new_$TAG.log
for i in 1 2; do echo "prev: $(timeout 200 python tools/bench_rows.py --lib libwanhip_prev.so 2>/dev/null | tr -d "\n " | cut -c1-400)"; echo "new:  $(timeout 200 python tools/bench_rows.py 2>/dev/null | tr -d "\n " | cut -c1-400)"; done | tee $O/${ROUND}_ab_rows_$TAG.log
for lib in libwanhip.so libwanhip_prev.so libwanhip.so; do echo "--- $lib"; (timeout 500 python tools/probes/dit_determinism.py A 100 big --lib $lib & timeout 500 python tools/probes/dit_determinism.py B 100 big --lib $lib; wait) 2>&1 | grep -E "forwards differ"; done | tee $O/${ROUND}_dit_determinism_rmsrope_wg_$TAG.log
(timeout 300 python tools/probes/dit_determinism.py CO 100 big > /dev/null 2>&1 & timeout 300 python tools/probes/rmsrope_twice.py V 50; wait) 2>&1 | grep -E "launches differ" | tee -a $O/${ROUND}_dit_determinism_rmsrope_wg_$TAG.log
