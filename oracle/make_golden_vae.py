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
    # spatial tiling (vae.py:676-717, :769-839, :841-881): tile 64 px on a 128 x 128 clip -> 3 x 3 overlapping tiles (8, 8 and 4
    # latents wide), seams blended over 16 px
    g = torch.Generator().manual_seed(22)
    zt = torch.randn(1, 16, 2, 16, 16, generator=g)
    vt = (torch.rand(1, 3, 5, 128, 128, generator=g) * 2 - 1)
    vt[:, :, 1:] *= 0.5
    with torch.no_grad():
        tdec = vae.spatial_tiled_decode(zt.clone(), scale, 64)
        # The tiled branch of decode_to_cpu_uint8 slices `latent_source` and moves the slice `.to(device, dtype)` before scaling
        # it IN PLACE (vae.py:799-804).  In the pipeline the source is on the CPU and the device is the GPU, so the slice is a
        # copy; run on the CPU in one dtype, `.to()` returns a VIEW and the overlapping latent columns get scaled twice.  A
        # float64 source (exact copy of the fp32 values) with the model dtype pinned to fp32 restores the copy.
        vae._model_dtype = torch.float32
        tu8 = vae.decode_to_cpu_uint8(zt.double(), scale, 64)
        tu8_crop = vae.decode_to_cpu_uint8(zt.double(), scale, 64, target_frames=3, target_height=100, target_width=120, frame_start=1)
        del vae._model_dtype
        tenc = vae.spatial_tiled_encode(vt.clone(), scale, 64)
    tiled = {"dec": tdec.numpy(), "dec_u8": tu8.numpy(), "dec_u8_crop": tu8_crop.numpy(), "enc": tenc.numpy(), "seed": np.array([22]),
             "tile_size": np.array([64])}
    np.savez_compressed(os.path.join(out_dir, "vae_tiled.npz"), **tiled)
    print("vae_tiled.npz", {k: v.shape for k, v in tiled.items()})
    print("vae_small.npz", {k: v.shape for k, v in out.items()}, "dec range", float(dec.min()), float(dec.max()),
          "u8 mean", float(u8.float().mean()))


def main_endframe(ns, out_dir):
    """any_end_frame (vae.py:590-606, :646-650): start + end image clips -- the last frame / latent frame bypasses the feature
    cache.  Encode of a 10-frame clip (1 + 4 + 4 + the end frame -> 4 latent frames) and decode of 4 latent frames
    (1 + 4 + 4 + 1 = 10 frames), float and uint8."""
    W = VO.synth_vae_weights()
    vae = build_ref_vae(ns, W)
    scale = VO.default_scale()
    g = torch.Generator().manual_seed(23)
    z = torch.randn(1, 16, 4, 8, 8, generator=g)
    vid = (torch.rand(1, 3, 10, 64, 64, generator=g) * 2 - 1)
    vid[:, :, 1:-1] *= 0.5
    with torch.no_grad():
        dec = vae.decode(z, scale, any_end_frame=True)
        u8 = vae.decode_to_cpu_uint8(z, scale, 0, any_end_frame=True)
        enc = vae.encode(vid, scale, any_end_frame=True)
    out = {"dec": dec.numpy(), "dec_u8": u8.numpy(), "enc": enc.numpy(), "seed": np.array([23])}
    np.savez_compressed(os.path.join(out_dir, "vae_endframe.npz"), **out)
    print("vae_endframe.npz", {k: v.shape for k, v in out.items()})
