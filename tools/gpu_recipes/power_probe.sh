# the MFMA power probe (tools/probes/mfma_power_probe.hip, built HERE before the call: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_power_probe tools/probes/mfma_power_probe.hip; the binary is git-ignored and travels), plain and with counters
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 120 tools/probes/mfma_power_probe 2>&1 | grep -v amdgpu.ids | tee $O/${ROUND}_mfma_power_probe_$TAG.log
cd /tmp
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/$O/pmc -o a --output-format csv -- $R/tools/probes/mfma_power_probe > $R/$O/pmc.log 2>&1
cd $R
python - "$O" "$ROUND" "$TAG" <<'PY'
import csv, glob, json, collections, sys
o, rnd, tag = sys.argv[1:4]
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + "/pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for f in glob.glob(o + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in dur.items():
    ms = sum(v) / len(v); c = {n: sum(x) / len(x) for n, x in cnt[k].items()}
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8
    out[k] = {"launches": len(v), "avg_ms": ms, "clock_GHz": g / ms / 1e6 if ms else None,
              "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else None, **c}
    if out[k]["mfma_util"]:
        print(f"{k[:48]:48s} {ms:8.2f} ms  clock {out[k]['clock_GHz']:.3f} GHz  mfma util {out[k]['mfma_util']:.3f}")
json.dump(out, open(f"{o}/{rnd}_mfma_power_probe_pmc_{tag}.json", "w"), indent=1)
PY
rm -rf $O/pmc
