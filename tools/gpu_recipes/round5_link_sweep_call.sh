# the link model's sensitivity: a world of 8 in the four layouts at 25 / 50 / 100 GB/s per peer, and the Ulysses rows with 2 and 5 head chunks
TAG=${TAG:-run05}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
SHORT="--steps 2 --warmup 1 --no-secondary --no-config5 --no-configs3 --no-cpu-baseline --no-robustness --no-e2e --simulate-world 8"
for cfg in "25 0" "100 0" "50 5"; do
  set -- $cfg
  ( timeout 400 python bench.py $SHORT --simulate-layout $( [ $2 = 0 ] && echo all || echo cfg-ulysses ) --simulate-link-GBs $1 --sp-chunks $2 ) > $O/${ROUND}_bench_simulated_world8_link$1_chunks$2_$TAG.json 2> $O/sim_$1_$2.err
  python - "$O/${ROUND}_bench_simulated_world8_link$1_chunks$2_$TAG.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "one-GPU ms/step", round(j["ms_per_step"], 1))
for row in j["simulated_scaling"]["ranks"]:
    oe = row.get("one_exchange", {})
    print("  %-24s compute %.3f link %.3f exposed %.2f ms/block | one exchange %.3f | %s" % (row["layout"], row["compute_side_efficiency"], row.get("link_modelled_efficiency", 0), row.get("exposed_ms_per_block", 0), oe.get("link_modelled_efficiency", 0), row["exchange"]))
PY
done 2>&1 | tee $O/${ROUND}_link_model_sweep_$TAG.log
