# a SHORT headline run on whatever box this call gets (boxes of the pool differ by up to 10 %: the bare MFMA loop in the line says which one it was)
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-config5 --no-configs3 --no-cpu-baseline --simulate-world "" ) > $O/${ROUND}_bench_14B-720p_short_$TAG.json 2> $O/bench_short.err; echo "bench rc=$?"
python - <<'P'
import json, os
p = os.path.join("gpurun_out", os.environ.get("TAG", "run"), os.environ.get("ROUND", "r04") + "_bench_14B-720p_short_" + os.environ.get("TAG", "run") + ".json")
j = json.load(open(p)); r = j["roofline"]
print("ms/step", round(j["ms_per_step"], 1), "attn TFLOP/s", round(r["achieved"], 1), "frac", round(r["frac"], 3), "sustained", round(r["sustained_mfma"]["TFLOPs"], 1), "of sustained", round(r["frac_of_sustained_mfma"], 3),
      "gain12", round(r.get("frac_gain_12", 0), 3), "others", {k: round(v, 1) for k, v in r["other_kernels"].items()}, "vae s", round(j["e2e"]["vae_decode_to_host_s"], 2))
P
