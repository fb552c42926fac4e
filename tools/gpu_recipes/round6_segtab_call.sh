# round 6: the segment-table walk of the multi-segment attention launch -- parity (segmented / sequence-parallel / Ulysses tests), then the
# launch-by-layout micro-benchmark (contiguous against the Ulysses rank's segments)
TAG=${TAG:-run04}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sp.py -q -x -k "attention or sp or ulysses or seg" -p no:cacheprovider ) > $O/${ROUND}_pytest_segtab_$TAG.log 2>&1; tail -8 $O/${ROUND}_pytest_segtab_$TAG.log
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" -p no:cacheprovider ) > $O/${ROUND}_pytest_gemm16s_$TAG.log 2>&1; tail -3 $O/${ROUND}_pytest_gemm16s_$TAG.log
timeout 500 python tools/bench_attn_shapes.py --heads 2,3,5 > $O/${ROUND}_attn_launch_by_layout_$TAG.log 2>&1; tail -4 $O/${ROUND}_attn_launch_by_layout_$TAG.log | cut -c1-600
