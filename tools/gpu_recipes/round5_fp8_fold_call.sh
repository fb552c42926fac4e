# the abs-max pass of four of a block's six fp8 activation quantisations folded into the producers (LayerNorms, ffn.0's GELU epilogue):
# parity (same rows, same scale, same fp8 bytes; the fp8 suites), the bf16 LayerNorm paths untouched (op / model suites), and what it buys on
# BASELINE configs[4]'s step -- folded / two-pass / folded on one box (WAN_FP8_NO_FOLD=1 = the two-pass form everywhere)
TAG=${TAG:-run08}; ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 500 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_fp8.py tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_mixed.py tests/test_gpu_skipcache.py ) > $O/${ROUND}_pytest_fp8_absmax_folded_$TAG.log 2>&1
echo "rc=$?" >> $O/${ROUND}_pytest_fp8_absmax_folded_$TAG.log; tail -3 $O/${ROUND}_pytest_fp8_absmax_folded_$TAG.log; grep -E "^(FAILED|ERROR)" $O/${ROUND}_pytest_fp8_absmax_folded_$TAG.log | head
SHORT="--workload i2v-14B-720p --fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --no-robustness --no-configs3 --no-config5 --simulate-world"
for leg in folded_a twopass folded_b; do
  NF=0; [ $leg = twopass ] && NF=1
  ( WAN_FP8_NO_FOLD=$NF timeout 400 python bench.py $SHORT "" ) > $O/${ROUND}_bench_config5_fp8_${leg}_$TAG.json 2> $O/bench_$leg.err
  python -c "import json; j=json.load(open('$O/${ROUND}_bench_config5_fp8_${leg}_$TAG.json')); print('$leg ms/step', round(j['ms_per_step'],1), 'ffn pair TF', round(j['roofline']['other_kernels'].get('ffn_gemm_pair_TFLOPs',0),1), 'sustained', round(j['roofline']['sustained_mfma']['TFLOPs'],1))"
done 2>&1 | tee $O/${ROUND}_ab_fp8_absmax_folded_config5_$TAG.log
