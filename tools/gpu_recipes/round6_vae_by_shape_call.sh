# round 6: the VAE decode's convolution launches one by one (rocprofv3 --kernel-trace): time per (kernel instantiation, grid) = per layer shape
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/$O/vae_kt -o a --output-format csv -- python $R/tools/bench_vae.py > $R/$O/vae_kt.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, sys, collections, json, re
f = glob.glob(sys.argv[1] + "/vae_kt/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"conv3d_halo_kernelILi(\d)ELb(\d)ELi(\d)ELi(\d)", n)
    if m: n = "halo<kt=%s,ups=%s,NB=%s,WCX=%s>" % m.groups()
    else: n = n.split("(")[0][-60:]
    g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?")
    k = (n, g)
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]
out = [{"kernel": k[0], "grid_x": k[1], "calls": v[0], "total_ms": round(v[1], 2), "pct": round(100 * v[1] / tot, 1), "avg_ms": round(v[1] / v[0], 4)} for k, v in rows]
json.dump({"label": "tools/bench_vae.py (two 720p x 81f decodes), per (kernel, grid)", "total_ms": round(tot, 1), "rows": out}, open(sys.argv[1] + "/r06_vae_decode_by_shape_%s.json" % sys.argv[1].split("/")[-1], "w"), indent=1)
for o in out: print(o)
PY
rm -rf $O/vae_kt
