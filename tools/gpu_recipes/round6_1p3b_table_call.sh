# round 6: BASELINE configs[1] (Wan2.1 t2v 1.3B 480x832x81f) under rocprofv3 --kernel-trace --stats on the round's final kernels: the per-kernel
# roofline table of the secondary workload (the 14B table is closing_sequence.sh's)
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_1p3b -o a --output-format csv -- python $R/bench.py --workload 1.3B-480p --steps 6 --warmup 2 --no-e2e --no-cpu-baseline > $R/$O/${ROUND}_bench_1.3B-480p_under_rocprofv3_$TAG.json 2> $R/$O/prof_1p3b.err
cd $R
python tools/rocprof_summarize.py $O/prof_1p3b $O/${ROUND}_1.3B-480p_kernel_trace_summary_$TAG.json "bench.py --workload 1.3B-480p --steps 6 --warmup 2 --no-e2e --no-cpu-baseline (8 CFG steps)" > /dev/null
python tools/roofline_table.py $O/${ROUND}_1.3B-480p_kernel_trace_summary_$TAG.json $O/${ROUND}_1.3B-480p_kernel_roofline_table_$TAG.json --workload 1.3B-480p > /dev/null
find $O/prof_1p3b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${ROUND}_1.3B-480p_rocprofv3_kernel_stats_$TAG.csv
rm -rf $O/prof_1p3b
head -c 400 $O/${ROUND}_bench_1.3B-480p_under_rocprofv3_$TAG.json; echo; tail -3 $O/prof_1p3b.err
python - $O/${ROUND}_1.3B-480p_kernel_roofline_table_$TAG.json <<'PY'
import json, sys
for k, v in json.load(open(sys.argv[1]))["lines"].items():
    print(k, v["achieved"], v["unit"], v["frac"], v["kernel"][:70])
PY
