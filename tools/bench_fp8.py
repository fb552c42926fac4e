#!/usr/bin/env python
"""fp8 vs bf16 GEMM at the Wan 14B shapes (tuning tool): M = 151,200 token rows."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:      # A/B against another build of the library (file name under wan2gp_amd/), e.g. libwanhip_f8k.so (make f8k)
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd import ops

M = 151200
res = {}
g = torch.Generator(device="cuda").manual_seed(0)
for name, K, N, epi in (("qkv", 5120, 5120, 0), ("ffn1_gelu", 5120, 13824, 1), ("ffn2", 13824, 5120, 0), ("o_gate", 5120, 5120, 2)):
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    ws = (w.abs().amax(dim=1) / 448).float()
    wq = (w / ws[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    wb = w.to(torch.bfloat16)
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    xq = ops.fp8_quantize(x)
    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fl = 2.0 * M * N * K
    kw = {}
    if epi == 2:
        kw = dict(residual=torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16), mod=torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16),
                  e=torch.randn(2, 6, N, device="cuda", generator=g).to(torch.bfloat16), gate_idx=5)
    ms8 = t(lambda: ops.linear_fp8(x, wq, ws, b, epilogue=epi, out=out, x_fp8=xq, **kw))
    msq = t(lambda: ops.fp8_quantize(x))
    ms16 = t(lambda: ops.linear(x, wb, b, epilogue=epi, out=out, **kw))
    res[name] = {"fp8_gemm_ms": ms8, "fp8_TF": fl / ms8 / 1e9, "quantize_ms": msq, "quantize_GBs": x.numel() * 5 / msq / 1e6,
                 "fp8_incl_quant_TF": fl / (ms8 + msq) / 1e9, "bf16_ms": ms16, "bf16_TF": fl / ms16 / 1e9}
    del x, w, wq, wb, out
print(json.dumps(res, indent=1))
