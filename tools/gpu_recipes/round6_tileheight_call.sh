# round 6: gemm256m at 160 / 192 / 224-row tiles -- parity (bit-identical to the 256-row tile), the in-process A/B, BASELINE configs[0] / [1] as timed workloads, configs[0] trace
TAG=${TAG:-run09}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" -p no:cacheprovider ) > $O/${ROUND}_pytest_gemm_tile_heights_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_gemm_tile_heights_$TAG.log
timeout 600 python tools/bench_gemm16s.py --set configs0,1.3B,rank8 > $O/${ROUND}_ab_gemm_tile_heights_$TAG.log 2>&1; grep -v "^{" $O/${ROUND}_ab_gemm_tile_heights_$TAG.log | tail -14 | cut -c1-900
for WL in 1.3B-320x512x17f 1.3B-480p; do
  timeout 600 python bench.py --workload $WL --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-secondary --simulate-world "" > $O/${ROUND}_bench_${WL}_$TAG.json 2> $O/bench_$WL.err; tail -2 $O/bench_$WL.err | head -1
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o a --output-format csv -- python $R/bench.py --workload 1.3B-320x512x17f --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-secondary --simulate-world "" > $R/$O/bench_c0_prof.json 2> $R/$O/bench_c0_prof.err
cd $R
python tools/rocprof_summarize.py $O/prof $O/${ROUND}_configs0_kernel_trace_summary_$TAG.json "bench.py --workload 1.3B-320x512x17f (35 CFG steps, replayed forwards)" > /dev/null; rm -rf $O/prof
python - <<'PY'
import json,os
d=json.load(open(os.path.join("gpurun_out",os.environ.get("TAG","run09"),os.environ.get("ROUND","r06")+"_configs0_kernel_trace_summary_"+os.environ.get("TAG","run09")+".json")))
tot=sum(v["total_ms"] for v in d["kernels"].values()); print("sum kernels ms per step", tot/35)
for k,v in list(d["kernels"].items())[:14]: print("%-50s %6d %9.1f %7.4f %5.1f%%"%(k[:50],v["calls"],v["total_ms"]/35,v["avg_ms"],v["pct"]))
PY
