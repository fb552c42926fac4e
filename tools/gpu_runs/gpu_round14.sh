#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 900 python tools/bench_attn.py --variants v2_8,abl8,abl16,abl32,abl24,abl48,abl56 --rounds 4 2>&1 | grep -E "TF_med|\"(v2_8|abl[0-9]+)\""
