#!/usr/bin/env python
"""Self-attention launch time by layout (round 6): the one-GPU form (B streams x L rows, K / V^T contiguous) against
the Ulysses rank's form (world x S query batches of L / world rows, K / V^T in `world` segments with ragged tails) at equal work,
for the head counts a rank of a world of 8 launches (2, 3, 5 of 40).  Per-workgroup-round time = ms / ceil(workgroups / CUs).
usage: bench_attn_shapes.py [--L 75600] [--world 8] [--heads 2,3,5,40] [--rounds 3]"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import ops  # noqa: E402


def timed(fn, rounds):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=75600)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--S", type=int, default=2)
    ap.add_argument("--heads", default="2,3,5,40")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    S, L, W = a.S, a.L, a.world
    Ll = L // W
    Lp = (Ll + 63) // 64 * 64
    cus = 256
    g = torch.Generator(device="cuda").manual_seed(0)
    res = []
    for H in [int(h) for h in a.heads.split(",")]:
        flops = 4.0 * S * H * L * L * 128
        # one-GPU form
        q = (torch.randn(S, L, H, 128, device="cuda", generator=g) * ops.attention_qscale()).to(torch.bfloat16)
        k = torch.randn(S, L, H, 128, device="cuda", generator=g).to(torch.bfloat16)
        vt = torch.randn(S, H * 128, (L + 63) // 64 * 64, device="cuda", generator=g).to(torch.bfloat16)
        scratch = torch.zeros(ops.attention_scratch_words(S, S, L, H), device="cuda")
        o = torch.empty_like(q)
        ms = timed(lambda: ops.attention(q, k, vt, Lk=L, out=o, q_prescaled=True, kmax_scratch=scratch), a.rounds)
        wgs = S * H * math.ceil(L / 256)
        row = {"heads": H, "contiguous": {"ms": ms, "TFLOPs": flops / ms / 1e9, "workgroups": wgs, "ms_per_round": ms / math.ceil(wgs / cus)}}
        del q, k, vt, o, scratch
        if H <= 10:
            # the Ulysses rank's form: q [world x S][Ll], K [seg][S][Ll][H 128], V^T [seg][S][H 128][Lp]
            q2 = (torch.randn(W * S, Ll, H, 128, device="cuda", generator=g) * ops.attention_qscale()).to(torch.bfloat16)
            k2 = torch.randn(W, S, Ll, H, 128, device="cuda", generator=g).to(torch.bfloat16)
            vt2 = torch.randn(W, S, H * 128, Lp, device="cuda", generator=g).to(torch.bfloat16)
            vt2[..., Ll:] = 0
            sc2 = torch.zeros(ops.attention_scratch_words(W * S, S, Ll, H), device="cuda")
            o2 = torch.empty_like(q2)
            ms2 = timed(lambda: ops.attention(q2, k2, vt2, Lk=Ll, out=o2, nseg=W, k_seg_stride=S * Ll * H * 128, vt_seg_stride=S * H * 128 * Lp,
                                              Bk=S, q_prescaled=True, kmax_scratch=sc2), a.rounds)
            wgs2 = W * S * H * math.ceil(Ll / 256)
            row["ulysses_segments"] = {"ms": ms2, "TFLOPs": flops / ms2 / 1e9, "workgroups": wgs2, "ms_per_round": ms2 / math.ceil(wgs2 / cus)}
            del q2, k2, vt2, o2, sc2
        res.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({"L": L, "world": W, "S": S, "rows": res}))


if __name__ == "__main__":
    main()
