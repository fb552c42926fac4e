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
a = ap.parse_args()
shapes = [("qkvo", a.M, 5120, 5120, 0), ("o+gate", a.M, 5120, 5120, 2), ("ffn1+gelu", a.M, 13824, 5120, 1), ("ffn2+gate", a.M, 5120, 13824, 2), ("vT", a.M // 2, 5120, 5120, 3)]
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
variants = [""]
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    mod = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    e = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    out = torch.empty(N, (M + 63) // 64 * 64, device="cuda", dtype=torch.bfloat16) if epi == 3 else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for vn in variants:
        ts = []
        for i in range(a.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.linear(x, w, b, epilogue=epi, residual=r, mod=mod, e=e, gate_idx=5 if epi == 2 else -1, out=out)
            e1.record(); torch.cuda.synchronize()
            if i: ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[f"{name}{':' + vn if vn else ''}"] = {"ms": ts[len(ts) // 2], "TF": 2.0 * M * N * K / ts[len(ts) // 2] / 1e9}
print(json.dumps(res, indent=1))
