# round 5, first GPU call: the parity tests of everything built on the CPU so far (advisor fixes, the chunked Ulysses exchange at small
# and at BASELINE sizes, the replayed launch list, the mixed plan at depth 40 + with the step-skipping caches), then three short benches:
#   a) the headline bf16 plan + the simulated world of 8 with the link model (chunked against one-exchange Ulysses rows),
#   b) the same steps with --mixed-precision (what the plan costs),
#   c) BASELINE configs[0] as the timed workload (replayed forwards), plain and under rocprofv3 --kernel-trace (step time against the
#      summed kernel time).
TAG=${TAG:-run01}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
T0=$(date +%s)
( time timeout 700 python -m pytest -q -s -m gpu -p no:cacheprovider tests/test_abi.py tests/test_gpu_model.py tests/test_gpu_skipcache.py tests/test_gpu_mixed.py \
    tests/test_gpu_14B_depth.py "tests/test_gpu_ops.py::test_rmsnorm_rope_persist_ragged_rows" "tests/test_gpu_ops.py::test_permute16_ex_pitched_column_ranges" \
    "tests/test_gpu_ops.py::test_permute16_is_a_block_transpose" tests/test_gpu_sp.py \
    "tests/test_gpu_baseline_configs.py::test_ulysses_world_rank_dryruns_at_baseline_size" ) > $O/${ROUND}_pytest_round5_new_$TAG.log 2>&1
echo "rc=$?" >> $O/${ROUND}_pytest_round5_new_$TAG.log; grep -E "passed|failed|error|rc=" $O/${ROUND}_pytest_round5_new_$TAG.log | tail -5; grep -E "^(FAILED|ERROR)" $O/${ROUND}_pytest_round5_new_$TAG.log | head -20
echo "tests took $(( $(date +%s) - T0 )) s"
SHORT="--steps 3 --warmup 1 --no-secondary --no-config5 --no-configs3 --no-cpu-baseline --no-robustness --no-e2e"
( timeout 500 python bench.py $SHORT --simulate-world 8 --simulate-layout all ) > $O/${ROUND}_bench_14B-720p_bf16_plan_link_model_$TAG.json 2> $O/bench_bf16.err; echo "bf16 plan rc=$?"
( timeout 400 python bench.py $SHORT --simulate-world "" --mixed-precision ) > $O/${ROUND}_bench_14B-720p_mixed_plan_$TAG.json 2> $O/bench_mixed.err; echo "mixed plan rc=$?"
( timeout 200 python bench.py --workload 1.3B-320x512x17f --steps 30 --warmup 5 --no-cpu-baseline --no-e2e ) > $O/${ROUND}_bench_configs0_replayed_$TAG.json 2> $O/bench_c0.err; echo "configs0 rc=$?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c0 -o a --output-format csv -- python $R/bench.py --workload 1.3B-320x512x17f --steps 30 --warmup 5 --no-cpu-baseline --no-e2e > $R/$O/${ROUND}_bench_configs0_under_rocprofv3_$TAG.json 2> $R/$O/prof_c0.err
cd $R
python tools/rocprof_summarize.py $O/prof_c0 $O/${ROUND}_configs0_kernel_trace_summary_$TAG.json "bench.py --workload 1.3B-320x512x17f --steps 30 --warmup 5 (35 CFG steps)" > /dev/null
rm -rf $O/prof_c0
python - "$O" "$ROUND" "$TAG" <<'PY'
import json, sys
o, r, t = sys.argv[1:4]
def ld(n):
    try:
        return json.load(open(f"{o}/{r}_{n}_{t}.json"))
    except Exception as ex:
        print("missing", n, ex)
        return None
a, b = ld("bench_14B-720p_bf16_plan_link_model"), ld("bench_14B-720p_mixed_plan")
if a and b:
    print("ms/step bf16 plan", round(a["ms_per_step"], 1), "mixed plan", round(b["ms_per_step"], 1), "ratio", round(b["ms_per_step"] / a["ms_per_step"], 4))
if a and "simulated_scaling" in a:
    for row in a["simulated_scaling"].get("ranks", []):
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in row.items() if k not in ("exchange",)})
c, d = ld("bench_configs0_replayed"), ld("bench_configs0_under_rocprofv3")
if c:
    print("configs0 ms/step", round(c["ms_per_step"], 2))
k = ld("configs0_kernel_trace_summary")
if k and d:
    tot = sum(v["total_ms"] for v in k["kernels"].values())
    print("configs0 under rocprofv3: ms/step", round(d["ms_per_step"], 2), "summed kernel ms per step", round(tot / 35.0, 2), "launches/step", sum(v["calls"] for v in k["kernels"].values()) / 35.0)
PY
echo "total $(( $(date +%s) - T0 )) s"
