# round 4: the VAE convolution's K loop with one stream removed at a time (make cabl): 1 = no gather DMA, 2 = no LDS reads / MFMAs, 3 = MFMAs on
# fragments read once.  Whole 720p x 81f decodes, alternating; outputs of the ablated builds are garbage, only their time is read.
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
for pass in 1 2; do for lib in libwanhip.so libwanhip_cabl1.so libwanhip_cabl2.so libwanhip_cabl3.so; do echo "== $lib" | tee -a $O/${ROUND}_vae_conv_ablation_$TAG.log; timeout 200 python tools/bench_vae.py --lib $lib 2>&1 | tail -1 | tee -a $O/${ROUND}_vae_conv_ablation_$TAG.log; done; done
