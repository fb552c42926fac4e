# the persistent GEMM (csrc/gemm256mp.hip) on the shapes it was written for: parity through the hook wan_gemm_debug_persist_max_k (references +
# bit identity against gemm256m.hip), then A/Bs inside ONE process (the two kernels alternate round by round): the Wan 1.3B projection shapes
# (K = 1536 / 8960), the 14B shapes again (K = 5120 / 13824: lost there in run 01), and the joint CFG forward of the 1.3B-480p workload
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "gemm" ) > $O/${ROUND}_pytest_gemm_persistent_short_k_$TAG.log 2>&1; echo "rc=$?" >> $O/${ROUND}_pytest_gemm_persistent_short_k_$TAG.log; tail -4 $O/${ROUND}_pytest_gemm_persistent_short_k_$TAG.log
timeout 200 python tools/bench_gemm.py --model 1.3B --ab-persist --rounds 9 > $O/${ROUND}_ab_gemm_persistent_1.3B_shapes_$TAG.log 2>&1; cat $O/${ROUND}_ab_gemm_persistent_1.3B_shapes_$TAG.log | tr -d '\n' | head -c 1500; echo
timeout 200 python tools/bench_gemm.py --model 14B --ab-persist --rounds 4 > $O/${ROUND}_ab_gemm_persistent_14B_shapes_$TAG.log 2>&1; cat $O/${ROUND}_ab_gemm_persistent_14B_shapes_$TAG.log | tr -d '\n' | head -c 1500; echo
timeout 200 python tools/bench_step_ab.py --workload 1.3B-480p --max-k 2048 --rounds 5 > $O/${ROUND}_ab_step_persistent_1.3B-480p_$TAG.log 2>&1; cat $O/${ROUND}_ab_step_persistent_1.3B-480p_$TAG.log | tr -d '\n' | head -c 1200; echo
timeout 200 python tools/bench_step_ab.py --workload 1.3B-480p --max-k 16384 --rounds 5 > $O/${ROUND}_ab_step_persistent_all_k_1.3B-480p_$TAG.log 2>&1; cat $O/${ROUND}_ab_step_persistent_all_k_1.3B-480p_$TAG.log | tr -d '\n' | head -c 1200; echo
