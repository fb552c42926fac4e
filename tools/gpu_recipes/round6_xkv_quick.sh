# round 6, tuning of attention_xkv.hip: its parity tests, the A/B against the persistent walk, the slot stamps of the tuning build (make xstamp)
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "kv_stationary" 2>&1 | tail -5 | tee $O/pytest_xkv_$TAG.log
timeout 300 python tools/bench_attn.py --L 75600 --Lk 512 --B 2 --H 40 --rounds 8 --modes bounded,persist 2>&1 | grep -E "TF_best|min_ms|bounded|persist" | tee $O/ab_xkv_$TAG.log
timeout 300 python tools/bench_attn.py --L 75600 --Lk 512 --B 2 --H 40 --rounds 3 --modes bounded --lib libwanhip_xstamp.so 2>&1 | grep -E "xkv stamps|min_ms" | head -9 | tee $O/stamps_xkv_$TAG.log
