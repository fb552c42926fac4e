#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC passes) into a small JSON/markdown
that can be committed under profiles/.   usage: rocprof_summarize.py <dir> <out.json> [label]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")      # (every kernel of the library lives in one; a name starting with "(" used to
    if name.startswith("void "):                            #  collapse to "" at the split below: round 5's mixed-plan trace lost its edge kernels)
        name = name[5:]
    name = name.split("(")[0]
    import re
    m = re.search(r"attn_w64q_kernel<(\d+)>|attn_w64q_kernelILi(\d+)E", name)
    if m:
        fl = int(m.group(1) or m.group(2))
        return "attn_w64q_kernel<%s%s>" % ("bounded" if fl & 4 else "tracking", ",prescaled" if fl & 2 else "")
    m = re.search(r"attn_w16n_kernel<(\d+)>|attn_w16n_kernelILi(\d+)E", name)
    if m:
        fl = int(m.group(1) or m.group(2))
        # round 4: the shifted twin (FLAGS | 128) is launched behind every plain launch and runs only the workgroups the plain one
        # handed over -- its own row, so that launches of the plain kernel still count forwards
        # (second half of round 4: FLAGS | 256 = the persistent short-KV form, text cross-attention)
        return "attn_w16n_kernel<%s%s>" % ("persistent" if fl & 256 else "shifted" if fl & 128 else "bounded", ",prescaled" if fl & 2 else "")
    m = re.search(r"conv3d_halo_kernelILi(\d)ELb(\d)", name)
    if m:
        return "conv3d_halo_kernel<kt=%s%s>" % (m.group(1), ",ups" if m.group(2) == "1" else "")
    if "conv3d_halo_kernel" in name:
        return "conv3d_halo_kernel"
    for k in ("gemm_fp8m_kernel", "gemm_fp8_kernel", "permute16_kernel", "fp8_quant_kernel", "fp8_absmax_kernel", "attn_kmax_kernel", "gemm256m_kernel", "gemm256k_kernel", "gemm256_kernel", "gemm32_kernel", "attn_w64q_kernel", "attn_w64_kernel", "attn_pp_kernel", "attn_fwd_kernel", "gemm_bf16_kernel", "rmsnorm_rope_kernel", "layernorm_kernel", "gated_residual",
              "patch_embed_kernel", "head_gemm_kernel", "gemv_kernel", "lincomb_kernel", "cfg_combine", "transpose_v"):
        if k in name:
            if k in ("gemm256k_kernel", "gemm256_kernel", "gemm32_kernel"):  # (gemm256m_kernel<EPI>: one template argument, kept under one name)
                import re
                m = re.search(k + r"ILi(\d)ELb(\d)", name)
                if m:
                    return f"{k}<epi={m.group(1)},bias_rows={m.group(2)}>"
            if k == "gemm_bf16_kernel":
                import re
                m = re.search(r"gemm_bf16_kernelILi(\d)ELb(\d)", name)
                if m:
                    return f"gemm_bf16_kernel<epi={m.group(1)},bias_rows={m.group(2)}>"
            return k
    return name[:60]


def main():
    d, out = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    res = {"label": label, "kernels": {}, "counters": {}}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        agg = defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            n = short(row.get("Kernel_Name", ""))
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
            agg[n][0] += 1
            agg[n][1] += dur
        tot = sum(v[1] for v in agg.values()) or 1.0
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            res["kernels"][n] = {"calls": c, "total_ms": round(t, 3), "avg_ms": round(t / c, 4), "pct": round(100 * t / tot, 2)}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for row in csv.DictReader(open(f)):
            n = short(row.get("Kernel_Name", ""))
            a = agg[n][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
        for n, cs in agg.items():
            for cn, (c, v) in cs.items():
                res["counters"].setdefault(n, {})[cn] = {"dispatches": c, "avg": v / c, "sum": v}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in list(res["kernels"].items())[:12]}, indent=1))
    for n, cs in res["counters"].items():
        print(n, {k: round(v["avg"], 1) for k, v in cs.items()})


if __name__ == "__main__":
    main()
