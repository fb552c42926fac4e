"""Diagnostics: is the tiled fp32 VAE decode bit-reproducible call after call (alone / beside another process on the same GPU)?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
vae = WanVAEHIP(state_dict=random_vae_state_dict())
g = torch.Generator().manual_seed(5)
z = torch.randn(16, 2, 16, 16, generator=g).cuda()
tag = sys.argv[1] if len(sys.argv) > 1 else "solo"
ref = vae.decode([z], 64)[0]
reft = [vae._decode_clip(z[:, :, i:i + 8, j:j + 8], False, True, False)[1] for i in (0, 6, 12) for j in (0, 6, 12)]
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    o = vae.decode([z], 64)[0]
    tiles = [vae._decode_clip(z[:, :, i:i + 8, j:j + 8], False, True, False)[1] for i in (0, 6, 12) for j in (0, 6, 12)]
    d = (o - ref).abs().max().item()
    dt = [((a - b).abs().max().item(), int((a != b).sum())) for a, b in zip(tiles, reft)]
    if d != 0 or any(x[0] != 0 for x in dt):
        bad += 1
        print(tag, "iteration", it, "decode maxdiff", d, "tiles", dt, flush=True)
print(tag, "done: %d differing iterations" % bad, flush=True)
