# round 4: text cross-attention (512 keys) as persistent workgroups -- parity (fp64, bit identity against one-block launches, in-place hand-over),
# the stations of one block (make pstamp), the three forms alternating at the bench shape (B 2, H 40, Lq 75,600, Lk 512), the attention suites
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "cross_attention_persistent" -x ) > $O/${ROUND}_pytest_cross_attention_persistent_$TAG.log 2>&1; tail -15 $O/${ROUND}_pytest_cross_attention_persistent_$TAG.log
[ -f wan2gp_amd/libwanhip_pstamp.so ] && timeout 300 python tools/bench_attn.py --Lk 512 --modes bounded --rounds 1 --lib libwanhip_pstamp.so 2>&1 | grep pstamps | tee $O/${ROUND}_cross_attention_block_stamps_$TAG.log
timeout 300 python tools/bench_attn.py --Lk 512 --modes bounded,oneblock,tracking --rounds 6 2>&1 | tee $O/${ROUND}_bench_cross_attention_$TAG.log | grep -E "TF_med|shape|min_ms"
timeout 300 python tools/bench_attn.py --Lk 512 --modes bounded,oneblock,tracking --rounds 6 --gain 8 2>&1 | tee $O/${ROUND}_bench_cross_attention_gain8_$TAG.log | grep -E "TF_med|shape"
[ -n "$SUITES" ] && { ( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_nag.py -q -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_ops_model_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_ops_model_$TAG.log; }
true
