"""s_memtime stamps of gemm256p.hip (tuning aid): needs wan2gp_amd/libwanhip_ptiming.so = the library with gemm_bf16.hip built
-DWAN_GEMM_PERSISTENT and gemm256p.hip built -DG256P_TIMING.  Workgroup 40, its third tile (and the first stage of its fourth)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from wan2gp_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "libwanhip_ptiming.so")
from wan2gp_amd import ops
L = lib.load()
L.wan_gemm256p_stamps.restype = ctypes.c_int
L.wan_gemm256p_stamps.argtypes = [ctypes.c_void_p]
for (M, N, K, epi) in ((151200, 5120, 5120, 0), (151200, 5120, 5120, 2), (151200, 13824, 5120, 1), (151200, 5120, 13824, 2)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    mod = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    e = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.linear(x, w, b, epilogue=epi, residual=r, mod=mod, e=e, gate_idx=5 if epi == 2 else -1, out=out); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        st = (ctypes.c_uint64 * 16)()
        L.wan_gemm256p_stamps(st)
        s = list(st)
        print(f"gemm256p M={M} N={N} K={K} epi={epi}: {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.0f} TF | zero+first stage {s[1] - s[0]}  loop {s[2] - s[0]}  "
              f"bias/gate setup {s[3] - s[2]}  convert+stores {s[4] - s[3]}  next tile's zero+first stage {s[5] - s[4]}  tile {s[4] - s[0]} cycles")
