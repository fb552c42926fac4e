"""Diagnostics for attention_xkv.hip: where (row mod 16, channel) its output differs from the persistent walk's."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wan2gp_amd import ops, lib as L

BF = torch.bfloat16
B, Lq, H = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2
g = torch.Generator().manual_seed(1)
q = torch.randn(B, Lq, H, 128, generator=g).to(BF).cuda(); k = torch.randn(B, 512, H, 128, generator=g).to(BF).cuda()
v = torch.randn(B, 512, H, 128, generator=g).to(BF).cuda()
qs = (q.float() * ops.attention_qscale()).to(BF)
vt = ops.transpose_v(v)
scratch = torch.zeros(ops.attention_scratch_words(B, B, Lq, H), device="cuda")
got = ops.attention(qs, k, vt, q_prescaled=True, kmax_scratch=scratch).float()
old = L.load().wan_attention_debug_no_persist(2)
walk = ops.attention(qs, k, vt, q_prescaled=True, kmax_scratch=scratch).float()
L.load().wan_attention_debug_no_persist(old)
bad = (got - walk).abs() > 4e-3          # [B, Lq, H, 128]
print("bad fraction", bad.float().mean().item(), "max", (got - walk).abs().max().item())
bd = bad[0].any(dim=1)                    # [Lq, 128]
print("bad channels:", torch.nonzero(bd.any(dim=0)).flatten().tolist())
rows = torch.nonzero(bd.any(dim=1)).flatten()
print("bad rows (count %d): first %s ... last %s" % (len(rows), rows[:24].tolist(), rows[-8:].tolist()))
print("bad rows mod 16:", sorted(set((rows % 16).tolist())))
print("bad tiles:", sorted(set((rows // 16).tolist()))[:40])
