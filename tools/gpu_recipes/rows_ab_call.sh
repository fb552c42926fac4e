# round 4: RMSNorm+RoPE with a third wave per SIMD (weights in LDS, scalar row pointers, FULL rows): bit identity against the previous
# build (libwanhip_rows_prev.so = HEAD's elementwise.hip, built by hand beside the library), the op tests, alternating timings
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
for lib in libwanhip_rows_prev.so libwanhip.so; do timeout 200 python tools/rows_hash.py --lib $lib 2>&1 | tail -1 | tee $O/${ROUND}_rows_hash_${lib%.so}_$TAG.log; done
cmp $O/${ROUND}_rows_hash_libwanhip_rows_prev_$TAG.log $O/${ROUND}_rows_hash_libwanhip_$TAG.log && echo "ROW KERNEL OUTPUTS BIT-IDENTICAL" | tee $O/${ROUND}_rows_bit_identity_$TAG.log
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "rmsnorm or layernorm or ln_" ) > $O/${ROUND}_pytest_rows_$TAG.log 2>&1; tail -3 $O/${ROUND}_pytest_rows_$TAG.log
for pass in 1 2 3; do for lib in libwanhip_rows_prev.so libwanhip.so; do echo "== $lib pass $pass"; timeout 200 python tools/bench_rows.py --lib $lib 2>&1 | grep -v amdgpu.ids | tee -a $O/${ROUND}_bench_rows_ab_$TAG.log | grep "rmsnorm"; done; done
