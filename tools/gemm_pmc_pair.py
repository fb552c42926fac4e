#!/usr/bin/env python
"""Counter pass helper (run under rocprofv3 --pmc ...): ONE Wan 14B projection shape, this library's GEMM and torch.addmm (hipBLASLt)
on the same tensors, `--n` launches each, alternating.  The per-kernel GRBM_GUI_ACTIVE over the kernel's duration is the clock the chip
ran at under each kernel; MFMA-busy cycles over that is the matrix-pipe utilisation.  Not used by the product."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="ffn1")
ap.add_argument("--M", type=int, default=151200)
ap.add_argument("--n", type=int, default=4)
a = ap.parse_args()
N, K = {"qkvo": (5120, 5120), "ffn1": (13824, 5120), "ffn2": (5120, 13824)}[a.shape]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(a.M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
o1 = torch.empty(a.M, N, device="cuda", dtype=torch.bfloat16)
o2 = torch.empty(a.M, N, device="cuda", dtype=torch.bfloat16)
for i in range(a.n):
    torch.addmm(b, x, w.t(), out=o1)
    ops.linear(x, w, b, epilogue=0, out=o2)
torch.cuda.synchronize()
print(a.shape, "max |vendor - ours|", (o1.float() - o2.float()).abs().max().item())
