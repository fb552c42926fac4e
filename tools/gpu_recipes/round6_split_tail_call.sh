# round 6: the split tail of the bounded self-attention launch -- parity, then A/B on one box: short headline benches (split on / off / on), the launch-by-layout micro-benchmark, the rank-of-8 trace
TAG=${TAG:-run11}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sp.py -q -x -k "attention or sp or ulysses or seg" -p no:cacheprovider ) > $O/${ROUND}_pytest_split_tail_$TAG.log 2>&1; tail -6 $O/${ROUND}_pytest_split_tail_$TAG.log
timeout 300 python tools/bench_attn_shapes.py --heads 2,3,5,40 > $O/${ROUND}_attn_launch_by_layout_split_$TAG.log 2>&1; grep '^{"heads' $O/${ROUND}_attn_launch_by_layout_split_$TAG.log | cut -c1-420
WAN_ATTN_SPLIT_TAIL=0 timeout 300 python tools/bench_attn_shapes.py --heads 2,3,5,40 > $O/${ROUND}_attn_launch_by_layout_whole_$TAG.log 2>&1; grep '^{"heads' $O/${ROUND}_attn_launch_by_layout_whole_$TAG.log | cut -c1-420
SHORT="--steps 3 --warmup 1 --no-e2e --no-secondary --no-cpu-baseline --no-config5 --no-configs3 --no-robustness --no-s1 --simulate-world 8 --simulate-layout ulysses --simulate-link-GBs 0"
for V in 1 0 1; do
  WAN_ATTN_SPLIT_TAIL=$V timeout 900 python bench.py $SHORT > $O/${ROUND}_bench_14B-720p_split_tail_${V}_$TAG.json 2> $O/bench_split_$V.err
  python - <<PY
import json
d=json.loads(open("$O/${ROUND}_bench_14B-720p_split_tail_${V}_$TAG.json").read().strip().split("\n")[-1])
r=d["simulated_scaling"]["ranks"][0]
print("split", $V, "ms/step", round(d["ms_per_step"],1), "attn avg ms", round(d["roofline"]["avg_ms"],2), "frac", round(d["roofline"]["frac"],4), "| rank step", round(r["rank_step_ms"],1), "eff", round(r["compute_side_efficiency"],4))
PY
done
