# round 6: RMSNorm + RoPE writing the Ulysses send layout (wan_rmsnorm_rope_pack) -- bit identity, the sequence-parallel suites, the rank-of-8 trace
TAG=${TAG:-run13}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sp.py tests/test_gpu_zzz_cfg_parallel.py -q -x -k "rmsnorm or sp or ulysses or cfg or pack" -p no:cacheprovider ) > $O/${ROUND}_pytest_pack_$TAG.log 2>&1; tail -6 $O/${ROUND}_pytest_pack_$TAG.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_rank -o a --output-format csv -- python $R/tools/rank_trace.py --world 8 --layout ulysses > $R/$O/rank_trace_stdout.json 2> $R/$O/rank_trace.err
cd $R
tail -2 $O/rank_trace.err
python tools/rank_trace_table.py $O/prof_rank $O/rank_trace_stdout.json $O/${ROUND}_rank_world8_kernel_trace_$TAG.json 8
rm -rf $O/prof_rank
