# round 4: the shifted bounded loop (attention_w16n.hip SHIFT): every attention / sequence-parallel test, the bench-shape mixed-loop test
# (-> gpurun_out/parity/*.json via the tests' _report), the micro-benchmark
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sp.py -q -m gpu -p no:cacheprovider -k "attention or sp" ) > $O/${ROUND}_pytest_attention_shift_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_attention_shift_$TAG.log
( timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu -s -p no:cacheprovider -k "attention or cfg4" ) > $O/${ROUND}_pytest_attention_bench_shape_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_attention_bench_shape_$TAG.log; grep "mixed loops" $O/${ROUND}_pytest_attention_bench_shape_$TAG.log | cut -c1-1500
timeout 300 python tools/bench_attn.py --rounds 4 2>&1 | tee $O/${ROUND}_bench_attn_$TAG.log | tail -20
