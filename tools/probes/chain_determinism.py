"""Diagnostics: producer -> gemm256m -> consumer chains launched again and again -- bit-reproducible beside another process on the same GPU?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wan2gp_amd import ops, lib as L
tag, iters, rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L.load().wan_gemm_debug_force_tile_rows(rows)
g = torch.Generator().manual_seed(2)
BF = torch.bfloat16
M, d, ffn = 3840, 512, 1536
x = torch.randn(2, M // 2, d, generator=g).to(BF).cuda()
mod = (torch.randn(6, d, generator=g) * 0.1).to(BF).cuda(); e = (torch.randn(2, 6, d, generator=g) * 0.1).to(BF).cuda()
w1 = (torch.randn(ffn, d, generator=g) * 0.05).to(BF).cuda(); b1 = torch.randn(ffn, generator=g).to(BF).cuda()
w2 = (torch.randn(d, ffn, generator=g) * 0.05).to(BF).cuda(); b2 = torch.randn(d, generator=g).to(BF).cuda()
wq = (torch.randn(d, d, generator=g) * 0.05).to(BF).cuda(); bq = torch.randn(d, generator=g).to(BF).cuda()
def ffn_chain():
    xm = ops.ln_modulate(x, mod, e, 3, 4)
    h = ops.linear(xm, w1, b1, epilogue=ops.EPI_GELU_TANH)
    xx = x.clone()
    return ops.linear(h, w2, b2, epilogue=ops.EPI_GATE_RES, residual=xx, mod=mod, e=e, gate_idx=5, out=xx)
def qkv_chain():
    xm = ops.ln_modulate(x, mod, e, 0, 1)
    q = ops.linear(xm, wq, bq); k = ops.linear(xm, wq, bq); vt = ops.linear(xm, wq, bq, epilogue=ops.EPI_TRANSPOSED)
    return torch.cat([q.flatten(), k.flatten(), vt.flatten()])
def ln_only():
    return ops.ln_modulate(x, mod, e, 0, 1)
for label, fn in (("ln_modulate alone", ln_only), ("ln -> ffn.0 (gelu) -> ffn.2 (gated residual in place)", ffn_chain), ("ln -> q, k, v^T", qkv_chain)):
    ref = fn().clone()
    bad = sum(int(not torch.equal(fn(), ref)) for _ in range(iters))
    print(tag, "rows", rows, label, ": %d of %d runs differ" % (bad, iters), flush=True)
