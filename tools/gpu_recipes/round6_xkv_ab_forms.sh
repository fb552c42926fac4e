# round 6: two forms of attention_xkv.hip against each other (LIBS = library files under wan2gp_amd/), alternating processes, with the persistent walk as the box's reference
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
for r in 1 2 3; do for lib in $LIBS; do
  echo "$lib: $(timeout 200 python tools/bench_attn.py --L 75600 --Lk 512 --B 2 --H 40 --rounds 6 --modes bounded,persist --lib $lib 2>&1 | grep -E 'min_ms' | tr -d '\n')"
done; done | tee $O/r06_xkv_ab_forms_$TAG.log
