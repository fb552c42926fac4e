# round 4: the scaled-fp8 GEMM on gemm256m's stage discipline and the K = 128 MFMA (gemm_fp8m.hip): the fp8 suite through the product
# library, then alternating timings against the previous kernel (libwanhip_f8k.so, `make -C wan2gp_amd/csrc f8k`)
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_fp8.py -q -m gpu -p no:cacheprovider ) > $O/${ROUND}_pytest_fp8m_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_fp8m_$TAG.log
for pass in 1 2; do for lib in libwanhip_f8k.so libwanhip.so; do echo "== $lib pass $pass"; timeout 300 python tools/bench_fp8.py --lib $lib 2>&1 | grep -v amdgpu.ids | tee $O/${ROUND}_bench_fp8_${lib%.so}_pass${pass}_$TAG.log | grep "fp8_TF\|bf16_TF"; done; done
