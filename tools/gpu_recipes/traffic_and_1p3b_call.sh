# (1) HBM-side traffic of the SHIPPED self-attention launch (attn_w16n_kernel, one bounded launch per call since round 4) at the bench
#     shape: FETCH_SIZE and WRITE_SIZE in passes of their own (--pmc with --kernel-trace only; MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950)
#     -> profiles/attn_pmc_traffic.json's source for `roofline.traffic`;
# (2) which memory-side counters this rocprofv3 offers at all (is there one that tells Infinity-Cache hits from HBM reads?);
# (3) BASELINE configs[1] (Wan2.1 t2v 1.3B 480x832x81f) under rocprofv3 --kernel-trace --stats: the per-kernel roofline table of the
#     secondary workload (the 14B table is closing_sequence.sh's).
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pmc_attn_f -o a --output-format csv -- python $R/tools/bench_attn.py --rounds 1 --modes bounded > $R/$O/pmc_attn_f.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pmc_attn_w -o a --output-format csv -- python $R/tools/bench_attn.py --rounds 1 --modes bounded > $R/$O/pmc_attn_w.log 2>&1
( timeout 60 rocprofv3-avail list 2>/dev/null || timeout 60 rocprofv3 -L 2>/dev/null ) | grep -i -E "mall|dram|hbm|TCC_EA0?_(RD|WR)REQ|TCC_(HIT|MISS|REQ)" | cut -c1-220 | sort -u | head -60 > $R/$O/${ROUND}_memory_side_counters_available_$TAG.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_1p3b -o a --output-format csv -- python $R/bench.py --workload 1.3B-480p --steps 6 --warmup 2 --no-e2e --no-cpu-baseline > $R/$O/${ROUND}_bench_1.3B-480p_under_rocprofv3_$TAG.json 2> $R/$O/prof_1p3b.err
cd $R
python tools/rocprof_summarize.py $O/pmc_attn_f $O/${ROUND}_14B_attn_w16n_one_launch_pmc_FETCH_SIZE_$TAG.json "the shipped self-attention launch at B=2 H=40 L=75600 (tools/bench_attn.py --rounds 1 --modes bounded), FETCH_SIZE pass (x2 on gfx950)" > /dev/null
python tools/rocprof_summarize.py $O/pmc_attn_w $O/${ROUND}_14B_attn_w16n_one_launch_pmc_WRITE_SIZE_$TAG.json "the same launch, WRITE_SIZE pass" > /dev/null
python tools/rocprof_summarize.py $O/prof_1p3b $O/${ROUND}_1.3B-480p_kernel_trace_summary_$TAG.json "bench.py --workload 1.3B-480p --steps 6 --warmup 2 --no-e2e --no-cpu-baseline (8 CFG steps)" > /dev/null
python tools/roofline_table.py $O/${ROUND}_1.3B-480p_kernel_trace_summary_$TAG.json $O/${ROUND}_1.3B-480p_kernel_roofline_table_$TAG.json --workload 1.3B-480p > /dev/null
find $O/prof_1p3b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${ROUND}_1.3B-480p_rocprofv3_kernel_stats_$TAG.csv
rm -rf $O/pmc_attn_f $O/pmc_attn_w $O/prof_1p3b
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_pmc_*SIZE*.json")):
    j = json.load(open(f))
    for k, v in j["kernels"].items():
        c = {n: x["avg"] for n, x in j["counters"].get(k, {}).items()}
        for n in ("FETCH_SIZE", "WRITE_SIZE"):
            if n in c and v["avg_ms"] > 0.2:
                print(f.split("/")[-1][:60], k[:50], v["calls"], round(v["avg_ms"], 2), n, "raw %.1f KB" % c[n])
PY
cat $O/${ROUND}_memory_side_counters_available_$TAG.txt | head -40
head -c 600 $O/${ROUND}_bench_1.3B-480p_under_rocprofv3_$TAG.json; echo; tail -3 $O/prof_1p3b.err; cat $O/${ROUND}_1.3B-480p_kernel_roofline_table_$TAG.json | head -70
