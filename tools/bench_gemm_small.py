#!/usr/bin/env python
"""GEMM micro-benchmark on the Wan 1.3B projection shapes (tuning tool)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import ops
M = 65520
shapes = [("qkvo", M, 1536, 1536, 0), ("ffn1+gelu", M, 8960, 1536, 1), ("ffn2+gate", M, 1536, 8960, 2), ("vT", M // 2, 1536, 1536, 3)]
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for name, m, n, k, epi in shapes:
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
    r = torch.randn(m, n, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    mod = torch.randn(1, 6, n, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    e = torch.randn(1, 6, n, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    out = torch.empty(n, (m + 63) // 64 * 64, device="cuda", dtype=torch.bfloat16) if epi == 3 else torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ts = []
    for i in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear(x, w, b, epilogue=epi, residual=r, mod=mod, e=e, gate_idx=5 if epi == 2 else -1, out=out)
        e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ts.sort()
    res[name] = round(2.0 * m * n * k / ts[len(ts) // 2] / 1e9, 1)
print(json.dumps(res))
