# the mixed-precision transformer plan: its parity suite, then what the plan COSTS at the headline shape -- a short
# bench of each plan on one box (bf16 plan first, then --mixed-precision), and a kernel trace of the mixed run for the wan_mx_* kernels' rates.
# (Round 4 ran only the parity suite, runs 37 / 38: the cost in DESIGN 3.12 is an estimate from the bytes until this has run.)
TAG=${TAG:-run}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 300 python -m pytest tests/test_gpu_mixed.py -q -s -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_mixed_precision_plan_$TAG.log 2>&1; tail -3 $O/${ROUND}_pytest_mixed_precision_plan_$TAG.log
SHORT="--steps 3 --warmup 1 --no-secondary --no-config5 --no-configs3 --no-cpu-baseline --no-robustness --simulate-world"
( timeout 500 python bench.py $SHORT "" ) > $O/${ROUND}_bench_14B-720p_bf16_plan_$TAG.json 2> $O/bench_bf16.err; echo "bf16 plan rc=$?"
( timeout 500 python bench.py $SHORT "" --mixed-precision ) > $O/${ROUND}_bench_14B-720p_mixed_plan_$TAG.json 2> $O/bench_mixed.err; echo "mixed plan rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_mx -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --no-robustness --simulate-world "" --mixed-precision > $R/$O/${ROUND}_bench_14B-720p_mixed_plan_under_rocprofv3_$TAG.json 2> $R/$O/prof_mx.err
cd $R
python tools/rocprof_summarize.py $O/prof_mx $O/${ROUND}_14B-720p_mixed_plan_kernel_trace_summary_$TAG.json "bench.py --mixed-precision --steps 1 --warmup 1 (2 CFG steps)" > /dev/null
rm -rf $O/prof_mx
python - "$O" "$ROUND" "$TAG" <<'PY'
import json, sys
o, r, t = sys.argv[1:4]
a = json.load(open(f"{o}/{r}_bench_14B-720p_bf16_plan_{t}.json")); b = json.load(open(f"{o}/{r}_bench_14B-720p_mixed_plan_{t}.json"))
print("ms/step bf16 plan", round(a["ms_per_step"], 1), "mixed plan", round(b["ms_per_step"], 1), "ratio", round(b["ms_per_step"] / a["ms_per_step"], 4))
k = json.load(open(f"{o}/{r}_14B-720p_mixed_plan_kernel_trace_summary_{t}.json"))["kernels"]
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
    print("%-70s %6d %10.2f ms" % (n[:70], v["calls"], v["total_ms"]))
PY
