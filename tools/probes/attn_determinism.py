"""Diagnostics: one attention call launched again and again on fixed operands -- bit-reproducible (and the same workgroups flagged) beside another
process on the same GPU?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wan2gp_amd import ops, lib as L
tag, iters = sys.argv[1], int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L.load().wan_attention_debug_no_persist(mode)
BF = torch.bfloat16
g = torch.Generator().manual_seed(4)
inplace = "--inplace" in sys.argv
for (B, Lq, Lk, H) in (((2, 14040, 14040, 12), (2, 1920, 1920, 4), (2, 6400, 6400, 12)) if inplace else ((2, 1920, 1920, 4), (2, 1920, 512, 4), (2, 6400, 6400, 12), (2, 6400, 512, 12))):
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF).cuda(); k = torch.randn(B, Lk, H, 128, generator=g).to(BF).cuda()
    v = torch.randn(B, Lk, H, 128, generator=g).to(BF).cuda()
    qs = (q.float() * ops.attention_qscale()).to(BF)
    vt = ops.transpose_v(v)
    scratch = torch.zeros(ops.attention_scratch_words(B, B, Lq, H), device="cuda")
    nflag = ((Lq + 255) // 256) * H * B
    def run():
        if not inplace:
            return ops.attention(qs, k, vt, q_prescaled=True, kmax_scratch=scratch)
        qq = qs.clone()
        return ops.attention(qq, k, vt, q_prescaled=True, kmax_scratch=scratch, out=qq)
    ref = run().clone()
    f0 = int((scratch[B * H:B * H + nflag].view(torch.int32) != 0).sum())
    bad = fl = 0
    for it in range(iters):
        o = run()
        f = int((scratch[B * H:B * H + nflag].view(torch.int32) != 0).sum())
        bad += int(not torch.equal(o, ref)); fl += int(f != f0)
    print(tag, "mode", mode, (B, Lq, Lk, H), ": %d of %d calls differ, flag count changed in %d (first call: %d flagged)" % (bad, iters, fl, f0), flush=True)
