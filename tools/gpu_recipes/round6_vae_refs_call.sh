# round 6: the VAE's causal caches as frame pointers (no copies) -- output hashes against the previous build, the VAE suites, A/B timing on
# one box (previous / new / previous / new), a kernel trace of the new build; then the s1 block + a fully measured video (bench --e2e-full)
TAG=${TAG:-run05}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( echo "prev:"; timeout 300 python tools/vae_hash.py --lib libwanhip_prev.so; echo "new:"; timeout 300 python tools/vae_hash.py ) > $O/${ROUND}_vae_hash_prev_vs_frame_pointers_$TAG.log 2>&1; cat $O/${ROUND}_vae_hash_prev_vs_frame_pointers_$TAG.log | grep -v amdgpu
( timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae22.py tests/test_gpu_vae_720p.py -q -x -p no:cacheprovider ) > $O/${ROUND}_pytest_vae_$TAG.log 2>&1; tail -4 $O/${ROUND}_pytest_vae_$TAG.log
for i in 1 2; do
  echo "prev:" >> $O/${ROUND}_ab_vae_frame_pointers_$TAG.log; timeout 300 python tools/bench_vae.py --encode --lib libwanhip_prev.so 2>/dev/null | tail -1 >> $O/${ROUND}_ab_vae_frame_pointers_$TAG.log
  echo "new:" >> $O/${ROUND}_ab_vae_frame_pointers_$TAG.log; timeout 300 python tools/bench_vae.py --encode 2>/dev/null | tail -1 >> $O/${ROUND}_ab_vae_frame_pointers_$TAG.log
done
cat $O/${ROUND}_ab_vae_frame_pointers_$TAG.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/vae_trace -o a --output-format csv -- python $R/tools/bench_vae.py --encode > $R/$O/vae_trace.log 2>&1
cd $R
python tools/rocprof_summarize.py $O/vae_trace $O/${ROUND}_vae_decode_encode_720p_kernel_trace_$TAG.json "tools/bench_vae.py --encode: two 720p x 81f decodes + one encode" | head -40
rm -rf $O/vae_trace
( time timeout 900 python bench.py --steps 1 --warmup 1 --no-secondary --no-robustness --no-configs3 --no-config5 --no-cpu-baseline --simulate-world "" --e2e-full ) > $O/${ROUND}_bench_14B-720p_s1_e2e_full_$TAG.json 2> $O/bench_e2e_full.err; tail -5 $O/bench_e2e_full.err
python - <<'PY'
import json,sys,os
d=json.loads(open(os.path.join("gpurun_out", os.environ.get("TAG","run05"), os.environ.get("ROUND","r06")+"_bench_14B-720p_s1_e2e_full_"+os.environ.get("TAG","run05")+".json")).read().strip().split("\n")[-1])
print(json.dumps({k:d.get(k) for k in ("ms_per_step","s1","e2e_full")}, indent=1)[:3000])
PY
