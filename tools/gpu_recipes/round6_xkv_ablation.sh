# round 6, attention_xkv.hip: what its iteration's time is made of -- the kernel with parts of its stream removed (make xabl), one process each
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
for n in 0 1 2 4 8 16 31; do
  lib=libwanhip_xabl$n.so; [ $n = 0 ] && lib=libwanhip.so
  echo "XKV_ABL=$n: $(timeout 200 python tools/bench_attn.py --L 75600 --Lk 512 --B 2 --H 40 --rounds 6 --modes bounded --lib $lib 2>&1 | grep -E 'min_ms' | head -1)"
done | tee $O/r06_xkv_ablation_$TAG.log
