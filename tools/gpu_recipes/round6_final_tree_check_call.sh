# round 6: after the aligned rotation -- the GPU test files that run elementwise.hip / mixed_ops.hip and were not in run 84, then a short bench line
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( time timeout 390 python -m pytest tests/test_gpu_14B_depth.py tests/test_gpu_baseline_configs.py tests/test_gpu_sp.py tests/test_gpu_fp8.py tests/test_gpu_skipcache.py tests/test_gpu_subparallel.py tests/test_gpu_vace_extra.py tests/test_gpu_zz_skip_layer_guidance.py tests/test_gpu_zzz_cfg_parallel.py tests/test_gpu_e2e.py -q -x -p no:cacheprovider ) 2>&1 | tail -6 | tee $O/${ROUND}_pytest_rest_of_dit_$TAG.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/${ROUND}_smoke_$TAG.log
timeout 170 python bench.py --steps 8 --warmup 2 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --no-robustness --no-s1 --simulate-world "" > $O/${ROUND}_bench_14B-720p_short_$TAG.json 2> $O/bench_short.err; echo "bench rc=$?"; head -c 700 $O/${ROUND}_bench_14B-720p_short_$TAG.json; echo
