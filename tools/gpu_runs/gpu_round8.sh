#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest e2e"
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25
echo "== vae 480p x 17f"
timeout 600 python tools/bench_vae.py --frames 17 --h 480 --w 832 --encode 2>&1 | tail -1
echo "== vae 720p x 81f"
timeout 900 python tools/bench_vae.py --encode 2>&1 | tee gpurun_out/bench_vae_720p.json | tail -1
echo "== rocprof vae 720p x 21f"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_vae -o v --output-format csv -- python $OLDPWD/tools/bench_vae.py --frames 21 ) > gpurun_out/rocprof_vae.log 2>&1
python tools/rocprof_summarize.py gpurun_out/prof_vae gpurun_out/r01_vae_kernel_trace.json "VAE decode 720p x 21f (x2)" | head -40
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
