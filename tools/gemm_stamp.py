"""s_memtime stamps of one stage of gemm256k.hip (tuning aid): needs wan2gp_amd/libwanhip_timing.so (`make -C wan2gp_amd/csrc timing`),
i.e. the library linked with gemm256k.hip compiled with -DG256K_TIMING (stage 30 of tile (0,0), wave 0; that tile's epilogue is skipped)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from wan2gp_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "libwanhip_timing.so")
from wan2gp_amd import ops
for (M, N, K) in ((151200, 5120, 5120), (151200, 5120, 13824)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.linear(x, w, b, out=out); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tl = out[1, :16].view(torch.int64).cpu().tolist()
        tiles_per_cu = ((M + 255) // 256) * ((N + 255) // 256) / 256
        tile_cycles = tl[3] - tl[0]
        print(f"  tile timeline (wg 40): prologue {tl[1] - tl[0]} loop {tl[2] - tl[1]} epilogue {tl[3] - tl[2]} total {tile_cycles} cycles; kernel {ms:.3f} ms = "
              f"{2.0 * M * N * K / ms / 1e9:.0f} TF; {tiles_per_cu:.1f} tiles per CU -> implied clock {tile_cycles * tiles_per_cu / ms / 1e6:.2f} GHz")
        st = out[0, :28].view(torch.int64).cpu().tolist()
        d = [st[i + 1] - st[i] for i in range(6)]
        print(f"gemm256k K={K} stage 30: k-step0 {d[0]} k-step1 {d[1]} k-step2 {d[2]} vmcnt/lgkm wait {d[3]} barrier {d[4]} k-step3 {d[5]} stage {st[6] - st[0]} (ideal 2048)")
