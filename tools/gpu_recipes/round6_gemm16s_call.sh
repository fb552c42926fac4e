# round 6: the small-tile GEMM (csrc/gemm16s.hip) -- parity on both tile heights, the other GEMM suites under the new dispatch, the in-process A/B
TAG=${TAG:-run03}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" -p no:cacheprovider ) > $O/${ROUND}_pytest_gemm16s_$TAG.log 2>&1; tail -15 $O/${ROUND}_pytest_gemm16s_$TAG.log
timeout 600 python tools/bench_gemm16s.py --set ${SETS:-all} > $O/${ROUND}_ab_gemm16s_$TAG.log 2>&1; grep -v "^{" $O/${ROUND}_ab_gemm16s_$TAG.log | tail -30
