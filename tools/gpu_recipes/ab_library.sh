# two builds of the library on ONE box, each alternating with torch.addmm (hipBLASLt) on the same tensors inside its process: the vendor
# kernel is the control that makes the two comparable (tools/gemm_vs_vendor.py --lib <file name under wan2gp_amd/>)
TAG=${TAG:-run}; ROUND=${ROUND:-r04}; LIB_A=${LIB_A:-libwanhip.so}; LIB_B=${LIB_B:-libwanhip_mp.so}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
for pass in 1 2; do for lib in $LIB_A $LIB_B; do
timeout 300 python tools/gemm_vs_vendor.py --lib $lib --rounds 5 2>&1 | tee $O/${ROUND}_gemm_vs_vendor_${lib%.so}_pass${pass}_$TAG.log | grep "ours_over_vendor_median\|library"
done; done
