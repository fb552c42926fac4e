# round 4, second half: ONE bounded launch per attention call (the shifted instantiation takes every workgroup) against the two-launch
# protocol (libwanhip_a2l.so, `make -C wan2gp_amd/csrc a2l`): the attention / SP / Ulysses suites on the product library, then
# alternating timings on the same tensors (gain 1: every row plain; mixed gains via the bench-shape test)
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sp.py -q -m gpu -p no:cacheprovider -k "attention or sp or ulysses" ) > $O/${ROUND}_pytest_attention_unified_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_attention_unified_$TAG.log
( timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu -s -p no:cacheprovider -k "attention or cfg4" ) > $O/${ROUND}_pytest_attention_bench_shape_unified_$TAG.log 2>&1; tail -3 $O/${ROUND}_pytest_attention_bench_shape_unified_$TAG.log; grep "mixed loops" $O/${ROUND}_pytest_attention_bench_shape_unified_$TAG.log | cut -c1-700
for pass in 1 2 3; do for lib in libwanhip_a2l.so libwanhip.so; do echo "== $lib pass $pass"; timeout 200 python tools/bench_attn.py --lib $lib --rounds 4 --modes bounded 2>&1 | grep "TF_med\|TF_best" | tee -a $O/${ROUND}_bench_attn_unified_ab_$TAG.log; done; done
