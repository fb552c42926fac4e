# kernel trace of two 720p x 81f decodes + one encode (tools/bench_vae.py --encode): where the VAE's time goes after the halo-patch kernel
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/vae_trace -o a --output-format csv -- python $R/tools/bench_vae.py --encode > $R/$O/vae_trace.log 2>&1
cd $R
python tools/rocprof_summarize.py $O/vae_trace $O/${ROUND}_vae_decode_encode_720p_kernel_trace_$TAG.json "tools/bench_vae.py --encode: two 720p x 81f decodes + one encode" | head -70
rm -rf $O/vae_trace
