#!/usr/bin/env python
"""Per-kernel roofline lines from a rocprofv3 kernel-trace summary of `bench.py --no-e2e --no-secondary --no-cpu-baseline`
(tools/rocprof_summarize.py output): algorithmic work per launch (SURVEY.md section 8d / DESIGN.md section 3) x launches
/ the kernel's total time in the trace, against the MI355X peaks (2.5 PFLOP/s dense bf16, 5 PFLOP/s fp8, 8 TB/s).
usage: roofline_table.py <summary.json> <out.json> [--workload 14B-720p] [--fp8]"""
import json
import sys

W = {"14B-720p": dict(S=2, L=75600, d=5120, ffn=13824, layers=40, text=512),
     "i2v-14B-720p": dict(S=2, L=75600, d=5120, ffn=13824, layers=40, text=512),
     "1.3B-480p": dict(S=2, L=32760, d=1536, ffn=8960, layers=30, text=512)}


def main():
    src, out = sys.argv[1], sys.argv[2]
    wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "14B-720p"
    fp8 = "--fp8" in sys.argv
    w = W[wl]
    S, L, d, ffn, text = w["S"], w["L"], w["d"], w["ffn"], w["text"]
    M = S * L
    k = json.load(open(src))["kernels"]

    def tot(prefix):                                            # a prefix or a tuple of prefixes
        rows = [(n, v) for n, v in k.items() if n.startswith(prefix)]
        return sum(v["calls"] for _, v in rows), sum(v["total_ms"] for _, v in rows)

    SELF = ("attn_w16n_kernel<bounded", "attn_w64q_kernel<bounded")      # the bounded loop: attention_w16n.hip since round 3 (16x16x32 MFMA)
    SELF_ALL = SELF + ("attn_w16n_kernel<shifted",)              # + its shifted twin (round 4): the time of both launches against one launch's work
    GEMM = ("gemm256m_kernel", "gemm256k_kernel")               # gemm256m.hip since round 3; gemm256k.hip keeps the row-bias (V^T) form
    n_self, _ = tot(SELF)
    if not n_self:                                              # one bounded launch per call since the second half of round 4: the shifted twin alone
        n_self, _ = tot("attn_w16n_kernel<shifted")
    forwards = n_self / w["layers"] if n_self else 0            # forward passes in the trace (one joint CFG pass each)
    # round 6: a self-attention call whose workgroups do not fill their last round of CUs is TWO launches of that name (the split tail's
    # finishing launch); the persistent cross-attention kernel is launched exactly once per block and forward
    n_cross, _ = tot("attn_xkv_kernel")              # round 6: the K / V^T-stationary kernel serves the text branch
    if not n_cross:
        n_cross, _ = tot("attn_w16n_kernel<persistent")
    if n_cross:
        forwards = n_cross / w["layers"]
    lines = {}

    def line(name, prefix, work_per_forward, unit, peak, note=""):
        calls, ms = tot(prefix)
        if not calls or not forwards:
            return
        ach = work_per_forward * forwards / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
        lines[name] = {"kernel": prefix if isinstance(prefix, str) else " + ".join(p for p in prefix if tot(p)[0]), "launches": calls, "total_ms": round(ms, 2), "achieved": round(ach, 1), "unit": unit,
                       "peak": peak, "frac": round(ach / peak, 3), "work_per_forward": work_per_forward, "note": note}

    # the bounded launch also serves cross-attention (short KV goes to the tracking instantiation): split by name
    line("self-attention", SELF_ALL, w["layers"] * 4.0 * S * L * L * d, "TFLOP/s", 2500.0, "4 S L^2 d per block; plain + shifted launch of every call")
    if "self-attention" in lines:
        lines["self-attention"]["launches"] = n_self
        lines["self-attention"]["calls"] = int(round(forwards * w["layers"]))
    line("cross-attention (Lk=512)", ("attn_xkv_kernel", "attn_w16n_kernel<persistent", "attn_w64q_kernel<tracking"), w["layers"] * 4.0 * S * L * text * d, "TFLOP/s", 2500.0,
         "4 S L 512 d per block; the K / V^T-stationary kernel (round 6; the persistent bounded walk of round 4 where a trace still has it) + every tracking launch of the trace (the hand-over passes behind self- and cross-attention: zero work)")
    big = w["layers"] * 2.0 * M * (6.0 * d * d + 2.0 * d * ffn)          # q,k,v,o, cross q,o, ffn1, ffn2 per block
    if fp8:
        line("GEMM (scaled fp8)", ("gemm_fp8m_kernel", "gemm_fp8_kernel"), big, "TFLOP/s", 5000.0, "2 M (6 d^2 + 2 d ffn) per block")
        line("fp8 activation quantisation", "fp8_", w["layers"] * 5.0 * M * (5.0 * d + 1.0 * ffn) , "GB/s", 8000.0,
             "absmax read 2 B + quantise read 2 B / write 1 B per element, 5 activations of width d and 1 of width ffn per block")
    else:
        line("GEMM (bf16)", GEMM, big, "TFLOP/s", 2500.0, "2 M (6 d^2 + 2 d ffn) per block")
    line("RMSNorm+RoPE", "rmsnorm_rope_kernel", w["layers"] * 6.0 * M * d * 2, "GB/s", 8000.0,
         "q,k (r+w) of self-attention + q (r+w) of cross-attention per block")
    line("LayerNorm family", "layernorm_kernel", (w["layers"] * 3 + 1) * 2.0 * M * d * 2, "GB/s", 8000.0, "norm1, norm2, norm3 per block + head")
    json.dump({"source": src, "workload": wl, "forwards_in_trace": forwards, "lines": lines}, open(out, "w"), indent=1)
    for n, v in lines.items():
        print(f"{n:32s} {v['achieved']:9.1f} {v['unit']:8s} frac {v['frac']:.3f}  ({v['launches']} launches, {v['total_ms']:.1f} ms)")


if __name__ == "__main__":
    main()
