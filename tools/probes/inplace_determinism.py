"""Diagnostics: is it IN-PLACE kernels that leave their first call's bits beside a neighbour that allocates / frees?  RMSNorm + RoPE in place
(wan_rmsnorm_rope) against the same arithmetic out of place (wan_rmsnorm_rope_pack, world 1), a torch in-place chain, the gated residual in
place -- interleaved, so that every form meets the same moments of the neighbour."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wan2gp_amd import ops
from oracle import wan_oracle as O
tag, iters = sys.argv[1], int(sys.argv[2])
BF = torch.bfloat16
g = torch.Generator().manual_seed(6)
d, grid = 1536, (9, 30, 52)
Lt = grid[0] * grid[1] * grid[2]
q0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
wq = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda()
y0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
mod = (torch.randn(6, d, generator=g) * 0.1).to(BF).cuda(); e = (torch.randn(2, 6, d, generator=g) * 0.1).to(BF).cuda()
cos, sin = [t.cuda() for t in O.rope_tables(grid)]
def inplace():
    q = q0.clone()
    ops.rmsnorm_rope_(q, None, wq, wq, freqs=(cos, sin), L=Lt, q_scale=ops.attention_qscale())
    return q
def outofplace():
    return ops.rmsnorm_rope_pack(q0, wq, 1, d // 128, freqs=(cos, sin), L=Lt, scale=ops.attention_qscale())
def torch_inplace():
    q = q0.clone()
    q.mul_(wq).add_(1.0)
    return q
def gated_inplace():
    x = q0.clone()
    ops.gated_residual_(x, y0, mod=mod, e=e, gate_idx=2)
    return x
forms = (("rmsnorm_rope in place", inplace), ("rmsnorm_rope_pack out of place", outofplace), ("torch mul_/add_ in place", torch_inplace), ("gated_residual in place", gated_inplace))
refs = [fn().clone() for _, fn in forms]
bad = [0] * len(forms)
for it in range(iters):
    for i, (_, fn) in enumerate(forms):
        bad[i] += int(not torch.equal(fn(), refs[i]))
for (label, _), b in zip(forms, bad):
    print(tag, label, ": %d of %d launches differ" % (b, iters), flush=True)
