#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
WAN_ATTN_VARIANT=w64f timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -p no:cacheprovider -k "attention" 2>&1 | tail -3
timeout 600 python tools/bench_attn.py --variants w64q,w64f --rounds 5 --stamps w64ft,w64qt 2>&1 | grep -E "stamps|TF_med|\"(w64q|w64f)\""
