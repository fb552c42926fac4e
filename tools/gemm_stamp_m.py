"""s_memtime stamps of gemm256m.hip (tuning aid): needs wan2gp_amd/libwanhip_mtiming.so (`make -C wan2gp_amd/csrc konly`): gemm256m.hip
built -DG256M_TIMING.  Workgroup 40: entry, end of the prologue, end of the main loop, end of the epilogue; and the clock the tile
cycles imply (cycles per tile x tiles per CU / kernel time)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from wan2gp_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "libwanhip_mtiming.so")
from wan2gp_amd import ops
L = lib.load()
L.wan_gemm256m_stamps.restype = ctypes.c_int
L.wan_gemm256m_stamps.argtypes = [ctypes.c_void_p]
cus = torch.cuda.get_device_properties(0).multi_processor_count
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
        L.wan_gemm256m_stamps(st)
        s = list(st)
        tiles = ((M + 255) // 256) * (N // 256)
        tile = s[3] - s[0]
        print(f"gemm256m M={M} N={N} K={K} epi={epi}: {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.0f} TF | prologue {s[1] - s[0]}  loop {s[2] - s[1]} "
              f"({(s[2] - s[1]) / (K // 64):.0f} per stage of 2048 MFMA cycles)  epilogue {s[3] - s[2]}  tile {tile} cycles; "
              f"pure MFMA {K // 64 * 2048} = {K // 64 * 2048 / tile:.3f} of the tile; implied clock {tile * tiles / cus / (ms * 1e-3) / 1e9:.3f} GHz")
