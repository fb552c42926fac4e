# full GPU suite, smoke, default bench.py, rocprofv3 kernel trace of a short bench (+ summary and per-kernel roofline table)
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( time timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_gpu_all_$TAG.log 2>&1; echo "rc=$?" >> $O/${ROUND}_pytest_gpu_all_$TAG.log; tail -6 $O/${ROUND}_pytest_gpu_all_$TAG.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/${ROUND}_smoke_$TAG.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/${ROUND}_bench_14B-720p_$TAG.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -4 $O/bench_default.err; head -c 1500 $O/${ROUND}_bench_14B-720p_$TAG.json; echo
cd /tmp
SHORT="--steps 2 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --simulate-world ''"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bench -o a --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --no-robustness --no-s1 --simulate-world "" > $R/$O/${ROUND}_bench_14B-720p_under_rocprofv3_$TAG.json 2> $R/$O/prof_bench.err
cd $R
python tools/rocprof_summarize.py $O/prof_bench $O/${ROUND}_14B-720p_kernel_trace_summary_$TAG.json "bench.py $SHORT (3 CFG steps)" > /dev/null
python tools/roofline_table.py $O/${ROUND}_14B-720p_kernel_trace_summary_$TAG.json $O/${ROUND}_14B-720p_kernel_roofline_table_$TAG.json > /dev/null
find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${ROUND}_14B-720p_rocprofv3_kernel_stats_$TAG.csv
rm -rf $O/prof_bench
head -c 400 $O/${ROUND}_bench_14B-720p_under_rocprofv3_$TAG.json; echo; head -60 $O/${ROUND}_14B-720p_kernel_roofline_table_$TAG.json
