"""Diagnostics: the row kernels (RMSNorm + RoPE, LayerNorm forms, gated residual, patch embedding, head) launched again and again on fixed
operands -- bit-reproducible beside another process on the same GPU?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wan2gp_amd import ops
from oracle import wan_oracle as O
tag, iters = sys.argv[1], int(sys.argv[2])
BF = torch.bfloat16
g = torch.Generator().manual_seed(6)
for d, H, grid in ((1536, 12, (9, 30, 52)), (512, 4, (5, 16, 24)), (5120, 40, (3, 30, 52))):
    F, Hg, Wg = grid
    Lt = F * Hg * Wg
    x = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
    q0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda(); k0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
    wq = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda(); wk = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda()
    cos, sin = [t.cuda() for t in O.rope_tables(grid)]
    mod = (torch.randn(6, d, generator=g) * 0.1).to(BF).cuda(); e = (torch.randn(2, 6, d, generator=g) * 0.1).to(BF).cuda()
    w3 = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda(); b3 = (0.1 * torch.randn(d, generator=g)).to(BF).cuda()
    y = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
    hmod = torch.randn(2, d, generator=g).cuda(); eh = torch.randn(2, d, generator=g).to(BF).cuda()   # (e: one row per stream)
    hw = (torch.randn(64, d, generator=g) * 0.05).cuda(); hb = torch.randn(64, generator=g).cuda()
    lat = torch.randn(2, 16, F, Hg * 2, Wg * 2, generator=g).cuda()
    pw = (torch.randn(d, 16, 1, 2, 2, generator=g) * 0.1).cuda(); pb = torch.randn(d, generator=g).cuda()
    def rr():
        q, k = q0.clone(), k0.clone()
        ops.rmsnorm_rope_(q, k, wq, wk, freqs=(cos, sin), L=Lt, q_scale=ops.attention_qscale())
        return torch.cat([q.flatten(), k.flatten()])
    def gr():
        xx = x.clone()
        ops.gated_residual_(xx, y, mod=mod, e=e, gate_idx=2)
        return xx
    forms = (("rmsnorm_rope(q, k)", rr), ("ln_modulate", lambda: ops.ln_modulate(x, mod, e, 0, 1)), ("ln_affine", lambda: ops.ln_affine(x, w3, b3)),
             ("gated_residual", gr), ("patch_embed", lambda: ops.patch_embed(lat, pw, pb)), ("head", lambda: ops.head(x, hmod, eh, hw, hb, grid)))
    for label, fn in forms:
        ref = fn().clone()
        bad = sum(int(not torch.equal(fn(), ref)) for _ in range(iters))
        print(tag, "d", d, "rows", 2 * Lt, label, ": %d of %d launches differ" % (bad, iters), flush=True)
