#!/usr/bin/env python
"""Reference point only: this library's GEMM against torch.addmm (hipBLASLt) on the Wan 14B projection shapes, IN ONE PROCESS on the
same tensors, launches alternating so that both see the same box, temperature and clock state (boxes of the pool differ by up to
15 % on GEMMs; a vendor figure from one run next to ours from another says nothing).  Each sample = `inner` back-to-back launches
between two events (the power-limited state is reached within the first few hundred ms), `rounds` samples per side, order swapped
every round.  Not used by the product."""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import lib as L
if "--lib" in sys.argv:      # another build of the library (file name under wan2gp_amd/)
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=151200)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--inner", type=int, default=8)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
shapes = [("q/k/v/o 5120x5120", a.M, 5120, 5120), ("ffn1 13824x5120", a.M, 13824, 5120), ("ffn2 5120x13824", a.M, 5120, 13824)]
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fn, inner):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner


def sustained():
    fl = ctypes.c_double(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.load().wan_mfma_sustained_probe(40000, ctypes.byref(fl), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    e1.record()
    torch.cuda.synchronize()
    return fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12


res = {"library": os.path.basename(L.LIB_PATH), "sustained_mfma_TFLOPs_before": sustained()}
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    o1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    vendor = lambda: torch.addmm(b, x, w.t(), out=o1)
    ours = lambda: ops.linear(x, w, b, epilogue=0, out=o2)
    timed(vendor, 3), timed(ours, 3)
    tv, to = [], []
    for r in range(a.rounds):
        if r & 1:
            to.append(timed(ours, a.inner)); tv.append(timed(vendor, a.inner))
        else:
            tv.append(timed(vendor, a.inner)); to.append(timed(ours, a.inner))
    tv.sort(); to.sort()
    fl = 2.0 * M * N * K / 1e9
    d = (o1.float() - o2.float()).abs().max().item()
    res[name] = {"vendor_TFLOPs_median": fl / tv[len(tv) // 2], "vendor_TFLOPs_best": fl / tv[0],
                 "ours_TFLOPs_median": fl / to[len(to) // 2], "ours_TFLOPs_best": fl / to[0],
                 "ours_over_vendor_median": tv[len(tv) // 2] / to[len(to) // 2], "max_abs_diff_of_the_two_outputs": d}
    del x, w, b, o1, o2
res["sustained_mfma_TFLOPs_after"] = sustained()
print(json.dumps(res, indent=1))
