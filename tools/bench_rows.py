#!/usr/bin/env python
"""Row-kernel micro-benchmark (tuning tool): RMSNorm+RoPE and LayerNorm+modulate at the 14B-720p and 1.3B-480p shapes.
Algorithmic bytes = one read + one write of every row; peak 8 TB/s.  --lib: another build to A/B."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for name, S, L, d in (("14B-720p", 2, 75600, 5120), ("1.3B-480p", 2, 32760, 1536)):
    q = torch.randn(S, L, d, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(S, L, d, device="cuda", generator=g).to(torch.bfloat16)
    wq = torch.randn(d, device="cuda", generator=g).to(torch.bfloat16)
    wk = torch.randn(d, device="cuda", generator=g).to(torch.bfloat16)
    cos = torch.randn(L, 128, device="cuda", generator=g)
    sin = torch.randn(L, 128, device="cuda", generator=g)
    mod = torch.randn(1, 6, d, device="cuda", generator=g).to(torch.bfloat16)
    e = torch.randn(1, 6, d, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty_like(q)
    runs = {"rmsnorm_rope": (lambda: ops.rmsnorm_rope_(q, k, wq, wk, (cos, sin), q_scale=0.1275), 4 * S * L * d * 2),
            "ln_modulate": (lambda: ops.ln_modulate(q, mod, e, 0, 1, out=out), 2 * S * L * d * 2)}
    for kn, (fn, nbytes) in runs.items():
        ts = []
        for i in range(a.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if i:
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        res[f"{kn} {name}"] = {"ms": round(ms, 4), "GBs": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3)}
print(json.dumps(res, indent=1))
