"""Diagnostics: one bf16 Linear launched again and again on fixed operands -- bit-reproducible beside another process on the same GPU?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "--lib" in sys.argv:
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd import ops, lib as L
tag, iters, rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L.load().wan_gemm_debug_force_tile_rows(rows)
g = torch.Generator().manual_seed(2)
BF = torch.bfloat16
shapes = ((3840, 1536, 512), (3840, 512, 1536), (3840, 512, 512), (6400, 1536, 1536), (37800, 5120, 5120))
if "--1p3b" in sys.argv:       # the Linears of the 1.3B model at 2 x 14,040 tokens (tools/probes/dit_determinism.py big)
    shapes = ((28080, 1536, 1536), (28080, 8960, 1536), (28080, 1536, 8960))
for (M, N, K) in shapes:
    x = torch.randn(M, K, generator=g).to(BF).cuda(); w = (torch.randn(N, K, generator=g) * 0.05).to(BF).cuda(); b = torch.randn(N, generator=g).to(BF).cuda()
    r = torch.randn(M, N, generator=g).to(BF).cuda()
    mod = torch.randn(6, N, generator=g).to(BF).cuda(); e = torch.randn(2, 6, N, generator=g).to(BF).cuda()
    rr = torch.empty_like(r)
    def inplace():
        rr.copy_(r)
        return ops.linear(x, w, b, epilogue=ops.EPI_GATE_RES, residual=rr, mod=mod, e=e, gate_idx=2, out=rr)
    def inplace_nogate():
        rr.copy_(r)
        return ops.linear(x, w, b, epilogue=ops.EPI_GATE_RES, residual=rr, out=rr)
    forms = (("vt", lambda: ops.linear(x, w, b, epilogue=ops.EPI_TRANSPOSED)), ("res_inplace_gated", inplace), ("res_inplace_no_gate", inplace_nogate)) if "--forms2" in sys.argv else \
            (("none", lambda: ops.linear(x, w, b)), ("gelu", lambda: ops.linear(x, w, b, epilogue=ops.EPI_GELU_TANH)),
             ("res", lambda: ops.linear(x, w, b, epilogue=ops.EPI_GATE_RES, residual=r)))
    for label, fn in forms:
        ref = fn().clone()
        n = iters if M < 20000 else max(20, iters // 4)
        bad = sum(int(not torch.equal(fn(), ref)) for _ in range(n))
        print(tag, "rows", rows, (M, N, K), label, ": %d of %d launches differ" % (bad, n), flush=True)
