#!/bin/bash
# One gpurun call: GPU parity tests, smoke, short benches.  Everything is logged under gpurun_out/.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== rocminfo"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4
echo "== pytest -m gpu"
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -60
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log | tail -5
for wl in "$@"; do
  echo "== bench $wl"
  timeout 900 python bench.py --workload $wl --steps 1 --warmup 1 2>&1 | tee gpurun_out/bench_$wl.log | tail -3
done
