#!/usr/bin/env python
"""sha256 of the VAE's outputs on fixed seeded inputs (tuning tool): run once per library build (--lib) and compare -- a kernel
rewrite that must not change a bit (round 3: the convolution's gather plan) is held against the previous build this way."""
import hashlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
vae = WanVAEHIP(state_dict=random_vae_state_dict())
res = {}
for name, (t, h, w) in (("128x160x13f", (4, 16, 20)), ("360x640x9f", (3, 45, 80)), ("720x1280x5f", (2, 90, 160))):
    g = torch.Generator().manual_seed(t * 1000 + h)
    z = torch.randn(16, t, h, w, generator=g)
    u8 = vae.decode_to_cpu_uint8([z], 0)[0]
    vid = torch.rand(3, (t - 1) * 4 + 1, h * 8, w * 8, generator=g) * 2 - 1
    mu = vae.encode([vid])[0].cpu()
    try:                                                    # (edge tiles whose latent h * w is not a multiple of 16 are refused by the attention block)
        tiled = vae.decode_to_cpu_uint8([z], 64)[0] if h <= 45 else None
    except Exception:
        tiled = None
    res[name] = {"decode": hashlib.sha256(u8.numpy().tobytes()).hexdigest()[:16], "encode": hashlib.sha256(mu.numpy().tobytes()).hexdigest()[:16],
                 "tiled": None if tiled is None else hashlib.sha256(tiled.numpy().tobytes()).hexdigest()[:16]}
print(json.dumps(res))
