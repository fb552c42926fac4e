#!/usr/bin/env python
"""GEMM micro-benchmark (tuning tool): the Wan 14B / 1.3B projection shapes, random bf16 data."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:      # A/B against another build of the library (file name under wan2gp_amd/)
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=151200)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--lib", default=None)
ap.add_argument("--model", default="14B", choices=["14B", "1.3B"], help="projection shapes: 14B (d 5120, ffn 13824, M = 2 x 75,600) or 1.3B (d 1536, ffn 8960, M = 2 x 32,760)")
a = ap.parse_args()
d, ffn = (5120, 13824) if a.model == "14B" else (1536, 8960)
if a.model == "1.3B" and a.M == 151200:
    a.M = 65520
shapes = [("qkvo", a.M, d, d, 0), ("o+gate", a.M, d, d, 2), ("ffn1+gelu", a.M, ffn, d, 1), ("ffn2+gate", a.M, d, ffn, 2), ("vT", a.M // 2, d, d, 3)]
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
variants = [""]
hook = None      # (round 4, run 32: a library hook alternated gemm256m.hip with the persistent form here, variants ["m", "mp"]; the form lost and is gone)
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    mod = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    e = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    out = torch.empty(N, (M + 63) // 64 * 64, device="cuda", dtype=torch.bfloat16) if epi == 3 else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tss = {vn: [] for vn in variants}
    for i in range(a.rounds + 1):                      # the variants alternate inside a round: same box, same clock state
        for vn in variants:
            if hook is not None:
                hook(0 if vn == "m" else 1 << 20)
            for rep in range(3):                       # three launches back to back per sample: short kernels (1.3B: 0.2 ms) settle
                if rep == 1:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ops.linear(x, w, b, epilogue=epi, residual=r, mod=mod, e=e, gate_idx=5 if epi == 2 else -1, out=out)
            e1.record(); torch.cuda.synchronize()
            if i: tss[vn].append(e0.elapsed_time(e1) / 2.0)
    for vn in variants:
        ts = sorted(tss[vn])
        res[f"{name}{':' + vn if vn else ''}"] = {"ms": ts[len(ts) // 2], "min_ms": ts[0], "TF": 2.0 * M * N * K / ts[len(ts) // 2] / 1e9}
if hook is not None:
    hook(0)
print(json.dumps(res, indent=1))
