#!/usr/bin/env python
"""sha256 of RMSNorm+RoPE / LayerNorm outputs on fixed seeded inputs (tuning tool): run once per library build (--lib) and compare --
a rewrite of a row kernel that must not change a bit (round 3: pairs; round 4: weights in LDS, scalar row pointers, a third wave per
SIMD) is held against the previous build this way.  Inputs come from the CPU generator, so two processes see the same bytes."""
import hashlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd import ops
res = {}
for name, S, L, d in (("14B", 2, 4801, 5120), ("1.3B", 2, 1333, 1536), ("5B", 1, 997, 3072), ("wide", 1, 515, 6144), ("ragged", 1, 301, 2560)):
    g = torch.Generator().manual_seed(d + L)
    q0 = torch.randn(S, L, d, generator=g).to(torch.bfloat16).cuda(); k0 = torch.randn(S, L, d, generator=g).to(torch.bfloat16).cuda()
    wq = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).cuda(); wk = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).cuda()
    ang = torch.rand(L, 128, generator=g) * 6.28
    cos, sin = torch.cos(ang).cuda(), torch.sin(ang).cuda()
    for tag, rope, qs in (("1.0", True, 1.0), ("0.1275", True, 0.1275), ("norope", False, 1.0)):
        q, k = q0.clone(), k0.clone()
        ops.rmsnorm_rope_(q, k, wq, wk, (cos, sin) if rope else None, q_scale=qs)
        torch.cuda.synchronize()
        res[f"{name}/{tag}"] = hashlib.sha256(q.cpu().view(torch.int16).numpy().tobytes() + k.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
print(json.dumps(res))
