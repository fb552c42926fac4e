# round 4: the fp32 plan of the Wan2.1 VAE (csrc/vae_f32.hip) against the reference's own fp32 CPU run (golden size) and the fp32 oracle at 720p
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_vae_720p.py -q -m gpu -s -p no:cacheprovider -k "fp32_plan" ) > $O/${ROUND}_pytest_vae_fp32_plan_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_vae_fp32_plan_$TAG.log; grep "VAE fp32 plan" $O/${ROUND}_pytest_vae_fp32_plan_$TAG.log | cut -c1-600
