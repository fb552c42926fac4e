# round 6: the text cache -- parity (model suites), the rank-of-8 trace again (segment table + text cache), BASELINE configs[0] and configs[1] as timed workloads
TAG=${TAG:-run07}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_nag.py tests/test_gpu_skipcache.py tests/test_gpu_fp8.py tests/test_gpu_mixed.py tests/test_gpu_vace_extra.py -q -x -p no:cacheprovider ) > $O/${ROUND}_pytest_textcache_$TAG.log 2>&1; tail -6 $O/${ROUND}_pytest_textcache_$TAG.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_rank -o a --output-format csv -- python $R/tools/rank_trace.py --world 8 --layout ulysses > $R/$O/rank_trace_stdout.json 2> $R/$O/rank_trace.err
cd $R
tail -2 $O/rank_trace.err
python tools/rank_trace_table.py $O/prof_rank $O/rank_trace_stdout.json $O/${ROUND}_rank_world8_kernel_trace_$TAG.json 8
rm -rf $O/prof_rank
for WL in 1.3B-320x512x17f 1.3B-480p; do
  timeout 600 python bench.py --workload $WL --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-secondary --simulate-world "" > $O/${ROUND}_bench_${WL}_$TAG.json 2> $O/bench_$WL.err; tail -2 $O/bench_$WL.err; head -c 700 $O/${ROUND}_bench_${WL}_$TAG.json; echo
done
