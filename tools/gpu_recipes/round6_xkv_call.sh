# round 6: text cross-attention with K / V^T stationary in registers (attention_xkv.hip): parity tests, then the A/B against the persistent walk
# (bounded = the product dispatch = xkv at 512 keys; persist = round 4's walk; oneblock = the same loop one block per workgroup), one process
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "cross_attention" 2>&1 | tail -25 | tee $O/${ROUND}_pytest_xkv_$TAG.log
timeout 300 python tools/bench_attn.py --L 75600 --Lk 512 --B 2 --H 40 --rounds 8 --modes bounded,persist,oneblock,tracking 2>&1 | tee $O/${ROUND}_ab_cross_attention_xkv_$TAG.log
