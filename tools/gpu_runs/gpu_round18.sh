#!/bin/bash
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 600 python tools/bench_attn.py --variants w64,w64p --rounds 3 --stamps w64t,w64pt 2>&1 | grep -E "stamps|TF_med|\"(w64|w64p)\""
