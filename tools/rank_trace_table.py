#!/usr/bin/env python
"""Cut a rocprofv3 kernel trace of tools/rank_trace.py at its marker launches and attribute a simulated rank's time over (one-GPU
step / world) to kernel classes.   usage: rank_trace_table.py <rocprof dir> <rank_trace stdout json> <out.json> [world]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summarize import short  # noqa: E402

CLASSES = (("self-attention", ("attn_w16n_kernel<shifted", "attn_w16n_kernel<bounded", "attn_w64q_kernel<bounded")),
           ("attention K pre-pass", ("attn_kmax_kernel",)),
           ("cross-attention + hand-over", ("attn_xkv_kernel", "attn_w16n_kernel<persistent", "attn_w64q_kernel<tracking")),
           ("GEMM tile kernels", ("gemm256m_kernel", "gemm256k_kernel", "gemm_fp8m_kernel")),
           ("GEMM small (gemm32 / first generation / gemv)", ("gemm32_kernel", "gemm_bf16_kernel", "gemv_kernel", "gemm_fp8_kernel", "gemm128")),
           ("RMSNorm + RoPE", ("rmsnorm_rope",)),
           ("LayerNorm family", ("layernorm_kernel",)),
           ("Ulysses re-packs", ("permute16",)),
           ("exchange stand-ins (copies)", ("__amd_rocclr_copyBuffer", "copyBuffer")),
           ("fills", ("__amd_rocclr_fillBuffer", "fillBuffer")),
           ("fp8 quantisation", ("fp8_",)))


def cls_of(name):
    for c, pre in CLASSES:
        if any(p in name for p in pre):
            return c
    return "other"


def main():
    d, js, out = sys.argv[1], sys.argv[2], sys.argv[3]
    world = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")),
                     r.get("Stream_Id", r.get("Queue_Id", ""))))
    rows.sort()
    # marker runs: consecutive delay_kernel launches
    phases, cur, run = {}, None, 0
    marks = []
    for i, r in enumerate(rows):
        if "delay_kernel" in r[2]:
            run += 1
        else:
            if run:
                marks.append((i, run))
            run = 0
    if run:
        marks.append((len(rows), run))
    idx = {n: i for i, n in marks}
    assert 3 in idx and 5 in idx and 7 in idx, "markers not found: %r" % marks
    phases["full"] = rows[idx[3]:idx[5] - 5]
    phases["rank"] = rows[idx[5]:idx[7] - 7]
    meta = json.loads([ln for ln in open(js) if ln.startswith("{")][-1])
    steps = {"full": meta["steps_per_phase"], "rank": meta["rank_phase_steps"]}
    agg = {p: defaultdict(lambda: [0, 0.0]) for p in phases}
    det = {p: defaultdict(lambda: [0, 0.0]) for p in phases}
    for p, rs in phases.items():
        for s, e, n, grid, _ in rs:
            nm = short(n)
            c = cls_of(nm if nm else n)
            agg[p][c][0] += 1
            agg[p][c][1] += (e - s) / 1e6
            k = "%s [grid %s]" % (nm, grid)
            det[p][k][0] += 1
            det[p][k][1] += (e - s) / 1e6
    table = {}
    tot_full = tot_rank = 0.0
    for c in sorted(set(agg["full"]) | set(agg["rank"])):
        fm = agg["full"][c][1] / steps["full"]
        rm = agg["rank"][c][1] / steps["rank"]
        tot_full += fm
        tot_rank += rm
        table[c] = {"one_gpu_ms_per_step": round(fm, 2), "ideal_rank_ms": round(fm / world, 2), "rank_ms_per_step": round(rm, 2),
                    "over_ideal_ms": round(rm - fm / world, 2), "launches_full": agg["full"][c][0] // steps["full"],
                    "launches_rank": agg["rank"][c][0] // steps["rank"]}
    ranks = meta["simulated"]["ranks"][0]
    res = {"source": "rocprofv3 --kernel-trace of tools/rank_trace.py, cut at its marker launches", "world": world, "layout": ranks.get("layout"),
           "one_gpu_step_ms_wall": meta["one_gpu_step_ms"], "rank_step_ms_wall": ranks.get("rank_step_ms"),
           "compute_side_efficiency": ranks.get("compute_side_efficiency"),
           "sum_kernels_one_gpu_ms": round(tot_full, 1), "sum_kernels_rank_ms": round(tot_rank, 1),
           "note": "rank kernels on the side stream (copies) overlap the compute stream: the sum of kernel times can exceed the wall step",
           "by_class": dict(sorted(table.items(), key=lambda kv: -kv[1]["over_ideal_ms"])),
           "rank_detail": {k: {"calls_per_step": v[0] / steps["rank"], "ms_per_step": round(v[1] / steps["rank"], 3), "avg_ms": round(v[1] / v[0], 4)}
                           for k, v in sorted(det["rank"].items(), key=lambda kv: -kv[1][1])[:40]},
           "full_detail": {k: {"calls_per_step": v[0] / steps["full"], "ms_per_step": round(v[1] / steps["full"], 3), "avg_ms": round(v[1] / v[0], 4)}
                           for k, v in sorted(det["full"].items(), key=lambda kv: -kv[1][1])[:25]}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("one_gpu_step_ms_wall", "rank_step_ms_wall", "compute_side_efficiency", "sum_kernels_one_gpu_ms", "sum_kernels_rank_ms")}))
    for c, v in res["by_class"].items():
        print("%-46s full %9.2f  ideal %8.2f  rank %8.2f  over %7.2f  (%d launches)" % (c, v["one_gpu_ms_per_step"], v["ideal_rank_ms"], v["rank_ms_per_step"], v["over_ideal_ms"], v["launches_rank"]))


if __name__ == "__main__":
    main()
