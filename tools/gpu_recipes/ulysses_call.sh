# round 4: Ulysses on hardware (all ranks on one GPU, gloo-staged exchanges), the ops it rests on, the whole attention / SP suite once
# more, and a SHORT bench with the four simulated layouts at a world of 8 (compute side of sp / cfg-sp / their Ulysses twins)
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sp.py tests/test_gpu_zzz_cfg_parallel.py -q -m gpu -p no:cacheprovider -k "attention or sp or ulysses or permute or cfg" ) > $O/${ROUND}_pytest_ulysses_$TAG.log 2>&1; tail -6 $O/${ROUND}_pytest_ulysses_$TAG.log
( timeout 900 python bench.py --steps 2 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --simulate-world 8 ) > $O/${ROUND}_bench_14B-720p_simulated_layouts_$TAG.json 2> $O/bench_sim.err; echo "bench rc=$?"; tail -3 $O/bench_sim.err
python - <<'P'
import json, os
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ.get("TAG", "run"), os.environ.get("ROUND", "r04") + "_bench_14B-720p_simulated_layouts_" + os.environ.get("TAG", "run") + ".json")
try:
    j = json.load(open(p))
    print("ms/step", j["ms_per_step"], "attn frac", j["roofline"]["frac"], "declined", j["roofline"]["declined_frac"])
    for r in j["simulated_scaling"]["ranks"]:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
except Exception as ex:
    print("no bench line:", ex)
P
