#!/usr/bin/env python
"""VAE decode/encode timing at full size (tuning tool)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:      # A/B against another build of the library (file name under wan2gp_amd/)
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=81); ap.add_argument("--h", type=int, default=720); ap.add_argument("--w", type=int, default=1280)
ap.add_argument("--encode", action="store_true"); ap.add_argument("--lib", default=None)
ap.add_argument("--no-halo", action="store_true", help="3x3x3 stride-1 convolutions on the gather kernel (wan_vae_debug_no_halo)")
a = ap.parse_args()
if a.no_halo:
    from wan2gp_amd import lib as _l2
    _l2.load().wan_vae_debug_no_halo(1)
vae = WanVAEHIP(state_dict=random_vae_state_dict())
t = (a.frames - 1) // 4 + 1
z = torch.randn(16, t, a.h // 8, a.w // 8, device="cuda")
res = {}
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u8 = vae.decode_to_cpu_uint8([z], 0)[0]
    torch.cuda.synchronize(); res[f"decode_s_{it}"] = time.perf_counter() - t0
res["out"] = list(u8.shape)
if a.encode:
    vid = torch.rand(3, a.frames, a.h, a.w, device="cuda") * 2 - 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mu = vae.encode([vid])[0]
    torch.cuda.synchronize(); res["encode_s"] = time.perf_counter() - t0
res["max_mem_GB"] = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps(res))
