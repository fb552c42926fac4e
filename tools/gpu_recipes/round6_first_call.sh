# round 6, run 01: (a) kernel trace of a simulated rank of a world of 8 (sp8 ulysses) beside the one-GPU step, cut at markers;
# (b) BASELINE configs[4] (scaled fp8) under rocprofv3 --kernel-trace + its per-kernel roofline table (there was none)
TAG=${TAG:-run01}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_rank -o a --output-format csv -- python $R/tools/rank_trace.py --world 8 --layout ulysses > $R/$O/rank_trace_stdout.json 2> $R/$O/rank_trace.err
cd $R
tail -3 $O/rank_trace.err
python tools/rank_trace_table.py $O/prof_rank $O/rank_trace_stdout.json $O/${ROUND}_rank_world8_kernel_trace_$TAG.json 8
rm -rf $O/prof_rank
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_fp8 -o a --output-format csv -- python $R/bench.py --workload i2v-14B-720p --fp8 --steps 2 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --no-robustness --simulate-world "" > $R/$O/${ROUND}_bench_config5_fp8_under_rocprofv3_$TAG.json 2> $R/$O/prof_fp8.err
cd $R
tail -3 $O/prof_fp8.err
python tools/rocprof_summarize.py $O/prof_fp8 $O/${ROUND}_config5_fp8_kernel_trace_summary_$TAG.json "bench.py --workload i2v-14B-720p --fp8 (3 CFG steps)" > /dev/null
python tools/roofline_table.py $O/${ROUND}_config5_fp8_kernel_trace_summary_$TAG.json $O/${ROUND}_config5_fp8_kernel_roofline_table_$TAG.json --workload i2v-14B-720p --fp8 > /dev/null
find $O/prof_fp8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${ROUND}_config5_fp8_rocprofv3_kernel_stats_$TAG.csv
rm -rf $O/prof_fp8
cat $O/${ROUND}_config5_fp8_kernel_roofline_table_$TAG.json | head -80
