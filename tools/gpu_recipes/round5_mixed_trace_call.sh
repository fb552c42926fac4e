# what the mixed-precision plan costs, kernel by kernel: the headline step under rocprofv3 --kernel-trace in both plans on ONE box
TAG=${TAG:-run03}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
SHORT="--steps 1 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --no-robustness --simulate-world"
cd /tmp
for plan in bf16 mixed; do
  X=""; [ $plan = mixed ] && X="--mixed-precision"
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$plan -o a --output-format csv -- python $R/bench.py $SHORT "" $X > $R/$O/${ROUND}_bench_14B-720p_${plan}_plan_under_rocprofv3_$TAG.json 2> $R/$O/prof_$plan.err
  ( cd $R; python tools/rocprof_summarize.py $O/prof_$plan $O/${ROUND}_14B-720p_${plan}_plan_kernel_trace_summary_$TAG.json "bench.py $SHORT '' $X (2 CFG steps)" > /dev/null; rm -rf $O/prof_$plan )
done
cd $R
python - "$O" "$ROUND" "$TAG" <<'PY'
import json, sys
o, r, t = sys.argv[1:4]
k = {p: json.load(open(f"{o}/{r}_14B-720p_{p}_plan_kernel_trace_summary_{t}.json"))["kernels"] for p in ("bf16", "mixed")}
names = sorted(set(k["bf16"]) | set(k["mixed"]), key=lambda n: -(k["mixed"].get(n, {}).get("total_ms", 0) + k["bf16"].get(n, {}).get("total_ms", 0)))
tb = sum(v["total_ms"] for v in k["bf16"].values()); tm = sum(v["total_ms"] for v in k["mixed"].values())
print("summed kernel ms (2 steps): bf16 %.1f mixed %.1f ratio %.4f" % (tb, tm, tm / tb))
for n in names[:22]:
    a, b = k["bf16"].get(n, {}), k["mixed"].get(n, {})
    print("%-62s bf16 %6d %9.2f | mixed %6d %9.2f | delta %+8.2f" % (n[:62], a.get("calls", 0), a.get("total_ms", 0), b.get("calls", 0), b.get("total_ms", 0), b.get("total_ms", 0) - a.get("total_ms", 0)))
PY
