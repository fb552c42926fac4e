#!/usr/bin/env python
"""Attention kernel micro-benchmark / A-B harness (tuning tool, not the judged bench).
Interleaved rounds of the variants in one process (cdna guide §5.4 rule 24), random data."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=75600)
    ap.add_argument("--Lk", type=int, default=0)
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--variants", default="base,lean,lean_prio,lean8,lean8_prio")
    ap.add_argument("--stamps", default="", help="variant that writes s_memtime stamps into O (w64t): print slot cycle deltas")
    a = ap.parse_args()
    Lk = a.Lk or a.L
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(a.B, a.L, a.H, 128, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(a.B, Lk, a.H, 128, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(a.B, Lk, a.H, 128, device="cuda", generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(v)
    variants = a.variants.split(",")
    for sv in (a.stamps.split(",") if a.stamps else []):
        os.environ["WAN_ATTN_VARIANT"] = sv
        for _ in range(2):
            o = ops.attention(q.clone(), k, vt)
            torch.cuda.synchronize()
            st = o.view(-1)[:80].view(torch.int64).cpu().tolist()
            if sv == "w64ft":   # fine stamps: [top, after barrier, tile start, then after gaps 3,7,...,67]
                print("stamps", sv, "top->barrier", st[1] - st[0], "per 4 gaps:", [st[i + 1] - st[i] for i in range(2, 19)],
                      "tile:", st[19] - st[0])
            else:
                print("stamps", sv, "deltas(top->barrier, A, B, C, D):", [st[i + 1] - st[i] for i in range(5)], "tile:", st[5] - st[0])
    flops = 4.0 * a.B * a.H * a.L * Lk * 128
    outs, times = {}, {vn: [] for vn in variants}
    for vn in variants:
        os.environ["WAN_ATTN_VARIANT"] = vn
        outs[vn] = ops.attention(q, k, vt).float()
    torch.cuda.synchronize()
    ref = outs[variants[0]]
    for r in range(a.rounds):
        for vn in variants:
            os.environ["WAN_ATTN_VARIANT"] = vn
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.attention(q, k, vt)
            e1.record()
            torch.cuda.synchronize()
            times[vn].append(e0.elapsed_time(e1))
    res = {}
    for vn in variants:
        t = sorted(times[vn])
        res[vn] = {"min_ms": t[0], "med_ms": t[len(t) // 2], "TF_best": flops / t[0] / 1e9, "TF_med": flops / t[len(t) // 2] / 1e9,
                   "maxdiff_vs_first": (outs[vn] - ref).abs().max().item()}
    print(json.dumps({"shape": [a.B, a.L, Lk, a.H], "results": res}, indent=1))


if __name__ == "__main__":
    main()
