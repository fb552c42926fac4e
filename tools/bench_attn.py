#!/usr/bin/env python
"""Attention kernel micro-benchmark / A-B harness (tuning tool, not the judged bench).
Interleaved rounds of the kernel's entry points in one process (cdna guide section 5.4 rule 24), random data:
  bounded    wan_attention_bounded with caller scratch: K pre-pass + bounded-softmax loop (the DiT's self-attention)
  tracking   the same kernel without a pre-pass: lazy-max loop
  generic    wan_attention on unscaled q (long KV: the 4x64 kernel with its pre-scaling pass; short KV: attn_pp<0,0,4>)
  oneblock   (short KV, --Lk 449..2048) `bounded` is then the persistent walk of round 4; this is the same loop, one q block per workgroup
  persist    (--Lk 512) `bounded` is then the K / V^T-stationary kernel of round 6 (attention_xkv.hip); this is round 4's persistent walk
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:      # A/B against another build of the library (file name under wan2gp_amd/)
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
elif "--stamps" in sys.argv:   # the stamp kernels live in the tuning build only (make -C wan2gp_amd/csrc timing)
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libwanhip_timing.so")
if "--stamps" in sys.argv:
    os.environ["WAN_ATTN_STAMPS"] = "1"
from wan2gp_amd import ops  # noqa: E402


def _one_block(fn, mode=1):
    """short KV (--Lk 449..2048): `bounded` is the persistent walk (512 keys: the K / V^T-stationary kernel); mode 1 = the same loop launched one
    q block per workgroup, mode 2 = the persistent walk"""
    from wan2gp_amd import lib as L
    old = L.load().wan_attention_debug_no_persist(mode)
    try:
        return fn()
    finally:
        L.load().wan_attention_debug_no_persist(old)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=75600)
    ap.add_argument("--Lk", type=int, default=0)
    ap.add_argument("--H", type=int, default=40)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--modes", default="bounded,tracking")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--gain", type=float, default=1.0, help="K scaled by this: beyond ~6 every workgroup of `bounded` takes the shifted loop (attention_w16n.hip SHIFT)")
    ap.add_argument("--stamps", action="store_true", help="library built with -DW64Q_TIMING and WAN_ATTN_STAMPS=1: print stamp deltas")
    a = ap.parse_args()
    Lk = a.Lk or a.L
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(a.B, a.L, a.H, 128, device="cuda", generator=g).to(torch.bfloat16)
    qs = (q.float() * ops.attention_qscale()).to(torch.bfloat16)
    k = (torch.randn(a.B, Lk, a.H, 128, device="cuda", generator=g) * a.gain).to(torch.bfloat16)
    v = torch.randn(a.B, Lk, a.H, 128, device="cuda", generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(v)
    del v
    scratch = torch.zeros(ops.attention_scratch_words(a.B, a.B, a.L, a.H), device="cuda")
    run = {"bounded": lambda: ops.attention(qs, k, vt, q_prescaled=True, kmax_scratch=scratch),
           "tracking": lambda: ops.attention(qs, k, vt, q_prescaled=True, kmax_scratch=False),
           "oneblock": lambda: _one_block(lambda: ops.attention(qs, k, vt, q_prescaled=True, kmax_scratch=scratch)),
           "persist": lambda: _one_block(lambda: ops.attention(qs, k, vt, q_prescaled=True, kmax_scratch=scratch), 2),
           "generic": lambda: ops.attention(q, k, vt)}
    modes = a.modes.split(",")
    if a.stamps:
        for m in modes:
            o = run[m]()
            torch.cuda.synchronize()
            st = o.view(-1)[:80].view(torch.int64).cpu().tolist()
            last = 18 if m == "bounded" else 19                      # 64 gaps (bounded) / 68 gaps (tracking)
            print("stamps", m, "top->barrier", st[1] - st[0], "barrier->tile", st[2] - st[1], "per 4 gaps:",
                  [st[i + 1] - st[i] for i in range(2, last)], "tile:", st[last] - st[0])
        return
    flops = 4.0 * a.B * a.H * a.L * Lk * 128
    outs, times = {}, {m: [] for m in modes}
    for m in modes:
        outs[m] = run[m]().float()
    torch.cuda.synchronize()
    ref = outs[modes[0]]
    for r in range(a.rounds):
        for m in modes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run[m]()
            e1.record()
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1))
    res = {}
    for m in modes:
        t = sorted(times[m])
        res[m] = {"min_ms": t[0], "med_ms": t[len(t) // 2], "TF_best": flops / t[0] / 1e9, "TF_med": flops / t[len(t) // 2] / 1e9,
                  "maxdiff_vs_first": (outs[m] - ref).abs().max().item()}
    print(json.dumps({"shape": [a.B, a.L, Lk, a.H], "results": res}, indent=1))


if __name__ == "__main__":
    main()
