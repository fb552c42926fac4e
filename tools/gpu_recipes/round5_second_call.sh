# round 5, second GPU call: what changed since the first -- the chunk-major Ulysses exchange (every tensor per head chunk), the fp32-stream
# GEMM epilogue and the register-resident fp32 LayerNorm of the mixed plan -- then the link-modelled table of worlds 2 / 4 / 8 in every
# layout (which layout `--parallelism auto` should take at each N) and what the mixed plan costs now.
TAG=${TAG:-run02}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
T0=$(date +%s)
( time timeout 600 python -m pytest -q -s -m gpu -p no:cacheprovider --durations=12 tests/test_gpu_mixed.py "tests/test_gpu_14B_depth.py::test_14B_forty_layers_vs_oracle[mixed_precision_plan]" \
    "tests/test_gpu_ops.py::test_rmsnorm_rope_persist_ragged_rows" tests/test_gpu_skipcache.py \
    "tests/test_gpu_sp.py::test_ulysses_chunked_exchange_is_bit_identical_to_one_exchange" "tests/test_gpu_sp.py::test_ulysses_ranks_at_baseline_size_vs_single_gpu_forward" \
    "tests/test_gpu_sp.py::test_ulysses_forward_ranks_on_one_gpu" "tests/test_gpu_zzz_cfg_parallel.py" \
    "tests/test_gpu_baseline_configs.py::test_ulysses_world_rank_dryruns_at_baseline_size" ) > $O/${ROUND}_pytest_round5_second_$TAG.log 2>&1
echo "rc=$?" >> $O/${ROUND}_pytest_round5_second_$TAG.log; grep -E "passed|failed|error|rc=" $O/${ROUND}_pytest_round5_second_$TAG.log | tail -5; grep -E "^(FAILED|ERROR)" $O/${ROUND}_pytest_round5_second_$TAG.log | head -20
echo "tests took $(( $(date +%s) - T0 )) s"
SHORT="--steps 3 --warmup 1 --no-secondary --no-config5 --no-configs3 --no-cpu-baseline --no-robustness --no-e2e"
( timeout 600 python bench.py $SHORT --simulate-world 2,4,8 --simulate-layout all ) > $O/${ROUND}_bench_14B-720p_bf16_plan_link_model_$TAG.json 2> $O/bench_bf16.err; echo "bf16 plan rc=$?"
( timeout 400 python bench.py $SHORT --simulate-world "" --mixed-precision ) > $O/${ROUND}_bench_14B-720p_mixed_plan_$TAG.json 2> $O/bench_mixed.err; echo "mixed plan rc=$?"
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
        oe = row.get("one_exchange", {})
        print(row.get("world"), "%-24s" % row.get("layout"), "compute", round(row.get("compute_side_efficiency", 0), 3), "link", round(row.get("link_modelled_efficiency", 0), 3),
              "exposed ms/block", round(row.get("exposed_ms_per_block", 0), 2), "| one exchange: link", round(oe.get("link_modelled_efficiency", 0), 3), "gain pts", round(row.get("chunking_gain_points", 0), 2))
PY
echo "total $(( $(date +%s) - T0 )) s"
