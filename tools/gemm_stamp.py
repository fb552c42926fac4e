import os, sys, torch
sys.path.insert(0, os.getcwd())
from wan2gp_amd import ops
M, N, K = 151200, 5120, 5120
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.linear(x, w, b, out=out)
    torch.cuda.synchronize()
    st = out[0, :20].view(torch.int64).cpu().tolist()
    print("gemm256 stamps (k-tile 60, wave 0): vmcnt wait", st[1] - st[0], "barrier", st[2] - st[1], "k-step 0 (16 MFMA + 4 DMA + 8 reads)", st[3] - st[2],
          "k-step 1", st[4] - st[3], "tile", st[4] - st[0])
