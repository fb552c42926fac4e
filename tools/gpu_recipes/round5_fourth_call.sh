# round 5, fourth GPU call: (a) how many 256 x 256 tiles a GEMM needs before the one-wave-per-SIMD kernel beats the 128-wide ones
# (WAN_GEMM_MIN_TILES sweep on BASELINE configs[0], where gemm32 is 56 % of a step; parity of the tile kernel on small / ragged problems
# with the threshold at 1); (b) the mixed plan against the bf16 plan A-B-A on one box, and its kernel trace after the edge-kernel change.
TAG=${TAG:-run04}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
T0=$(date +%s)
( WAN_GEMM_MIN_TILES=1 timeout 400 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_t5.py tests/test_gpu_mixed.py ) > $O/${ROUND}_pytest_gemm256m_on_small_problems_$TAG.log 2>&1
echo "rc=$?" >> $O/${ROUND}_pytest_gemm256m_on_small_problems_$TAG.log; tail -3 $O/${ROUND}_pytest_gemm256m_on_small_problems_$TAG.log; grep -E "^(FAILED|ERROR)" $O/${ROUND}_pytest_gemm256m_on_small_problems_$TAG.log | head
for mt in 256 128 64 32 256 64; do
  ( WAN_GEMM_MIN_TILES=$mt timeout 200 python bench.py --workload 1.3B-320x512x17f --steps 40 --warmup 5 --no-cpu-baseline --no-e2e ) > $O/c0_$mt.json 2> $O/c0_$mt.err
  python -c "import json,sys; j=json.load(open('$O/c0_$mt.json')); print('configs0 min_tiles $mt ms/step', round(j['ms_per_step'],2), j.get('forwards',{}).get('last_forward_how'))"
done 2>&1 | tee $O/${ROUND}_ab_gemm_min_tiles_configs0_$TAG.log
for mt in 256 64; do
  ( WAN_GEMM_MIN_TILES=$mt timeout 300 python bench.py --workload 1.3B-480p --steps 6 --warmup 2 --no-cpu-baseline --no-e2e ) > $O/c1_$mt.json 2> $O/c1_$mt.err
  python -c "import json,sys; j=json.load(open('$O/c1_$mt.json')); print('configs1 1.3B-480p min_tiles $mt ms/step', round(j['ms_per_step'],2))"
done 2>&1 | tee -a $O/${ROUND}_ab_gemm_min_tiles_configs0_$TAG.log
echo "gemm part took $(( $(date +%s) - T0 )) s"
SHORT="--steps 3 --warmup 1 --no-secondary --no-config5 --no-configs3 --no-cpu-baseline --no-robustness --no-e2e --simulate-world"
for leg in bf16_a mixed bf16_b; do
  X=""; [ $leg = mixed ] && X="--mixed-precision"
  ( timeout 400 python bench.py $SHORT "" $X ) > $O/${ROUND}_bench_14B-720p_${leg}_$TAG.json 2> $O/bench_$leg.err
  python -c "import json; j=json.load(open('$O/${ROUND}_bench_14B-720p_${leg}_$TAG.json')); print('$leg ms/step', round(j['ms_per_step'],1), 'sustained', round(j['roofline']['sustained_mfma']['TFLOPs'],1))"
done 2>&1 | tee $O/${ROUND}_ab_mixed_plan_cost_$TAG.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_mixed -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --no-robustness --simulate-world "" --mixed-precision > $R/$O/${ROUND}_bench_14B-720p_mixed_plan_under_rocprofv3_$TAG.json 2> $R/$O/prof_mixed.err
cd $R
python tools/rocprof_summarize.py $O/prof_mixed $O/${ROUND}_14B-720p_mixed_plan_kernel_trace_summary_$TAG.json "bench.py --mixed-precision --steps 1 --warmup 1 (2 CFG steps)" > /dev/null; rm -rf $O/prof_mixed
python - "$O" "$ROUND" "$TAG" <<'PY'
import json, sys
o, r, t = sys.argv[1:4]
k = json.load(open(f"{o}/{r}_14B-720p_mixed_plan_kernel_trace_summary_{t}.json"))["kernels"]
print("summed kernel ms (2 steps, mixed): %.1f" % sum(v["total_ms"] for v in k.values()))
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["total_ms"])[:18]:
    print("%-64s %6d %9.2f ms" % (n[:64], v["calls"], v["total_ms"]))
PY
echo "total $(( $(date +%s) - T0 )) s"
