"""TEST INFRASTRUCTURE ONLY -- VAE goldens from the REFERENCE's own WanVAE_ (vae.py), fp32 CPU.
Called by oracle/make_golden.py (needs /root/reference)."""
import os

import numpy as np
import torch

from oracle import vae_oracle as VO


def build_ref_vae(ns, W):
    vae = ns.V.WanVAE_(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                       temperal_downsample=[False, True, True]).eval()
    missing, unexpected = vae.load_state_dict({k: v.clone() for k, v in W.items()}, strict=True)
    return vae


def main(ns, out_dir):
    W = VO.synth_vae_weights()
    vae = build_ref_vae(ns, W)
    scale = VO.default_scale()
    g = torch.Generator().manual_seed(21)
    z = torch.randn(1, 16, 3, 8, 8, generator=g)
    with torch.no_grad():
        dec = vae.decode(z, scale)
        u8 = vae.decode_to_cpu_uint8(z, scale, 0)
        vid = (torch.rand(1, 3, 9, 64, 64, generator=g) * 2 - 1)
        vid[:, :, 1:] *= 0.5
        enc = vae.encode(vid, scale)
    out = {"dec": dec.numpy(), "dec_u8": u8.numpy(), "enc": enc.numpy(), "seed": np.array([21])}
    np.savez_compressed(os.path.join(out_dir, "vae_small.npz"), **out)
    print("vae_small.npz", {k: v.shape for k, v in out.items()}, "dec range", float(dec.min()), float(dec.max()),
          "u8 mean", float(u8.float().mean()))
