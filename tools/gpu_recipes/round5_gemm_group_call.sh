# the tile order of gemm256m at the short-K (Wan 1.3B) shapes: y tiles per group swept (WAN_GEMM_GROUP; 4 ships) -- does a different
# rasterisation cut the operand stream across the L2's fabric side (3.4 x algorithmic at K = 1,536, round 4 run 33) enough to show in time?
TAG=${TAG:-run07}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
for g in 4 1 2 8 16 32 64 4; do
  WAN_GEMM_GROUP=$g timeout 120 python tools/bench_gemm.py --model 1.3B --rounds 6 > $O/g13_$g.json 2> $O/g13_$g.err
  python -c "import json; j=json.load(open('$O/g13_$g.json')); print('1.3B group $g', {k: round(v['TF'], 1) for k, v in j.items()})"
done 2>&1 | tee $O/${ROUND}_ab_gemm_tile_order_group_$TAG.log
for g in 4 2 8 16; do
  WAN_GEMM_GROUP=$g timeout 120 python tools/bench_gemm.py --model 14B --rounds 3 > $O/g14_$g.json 2> $O/g14_$g.err
  python -c "import json; j=json.load(open('$O/g14_$g.json')); print('14B group $g', {k: round(v['TF'], 1) for k, v in j.items()})"
done 2>&1 | tee -a $O/${ROUND}_ab_gemm_tile_order_group_$TAG.log
