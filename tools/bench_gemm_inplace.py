#!/usr/bin/env python
"""ffn2 / o-projection with the residual updated in place (out is R), as wan_dit_forward calls them."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import ops
M = 151200
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for name, n, k in (("o+gate", 5120, 5120), ("ffn2+gate", 5120, 13824)):
    x = torch.randn(M, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
    mod = torch.randn(1, 6, n, device="cuda", generator=g).to(torch.bfloat16)
    for nb in (1, 2):
        e = torch.randn(nb, 6, n, device="cuda", generator=g).to(torch.bfloat16)
        for inplace in (False, True):
            r = torch.randn(M, n, device="cuda", generator=g).to(torch.bfloat16)
            out = r if inplace else torch.empty_like(r)
            xx = x.view(nb, M // nb, k)
            ts = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.linear(xx, w, b, epilogue=2, residual=r.view(nb, M // nb, n), mod=mod, e=e, gate_idx=5, out=out.view(nb, M // nb, n))
                e1.record(); torch.cuda.synchronize()
                if i: ts.append(e0.elapsed_time(e1))
            ts.sort()
            res[f"{name} batches={nb} inplace={inplace}"] = round(2.0 * M * n * k / ts[len(ts) // 2] / 1e9, 1)
print(json.dumps(res, indent=0))
