#!/usr/bin/env python
"""Reference point only: torch.matmul (hipBLASLt / rocBLAS) on the Wan 14B projection shapes, random bf16 data.
Not used by the product; tells how far the hand-written GEMM is from the vendor library on the same box."""
import json, torch
M = 151200
shapes = [("qkvo", M, 5120, 5120), ("ffn1", M, 13824, 5120), ("ffn2", M, 5120, 13824)]
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for name, m, n, k in shapes:
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.addmm(b, x, w.t(), out=out)
        e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ts.sort()
    res[name] = {"ms": ts[len(ts) // 2], "TF": 2.0 * m * n * k / ts[len(ts) // 2] / 1e9}
print(json.dumps(res))
