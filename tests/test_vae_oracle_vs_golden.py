"""Pins oracle/vae_oracle.py to the reference-generated VAE fixture (CPU, fp32, bit-exact)."""
import os

import numpy as np
import torch

from oracle import vae_oracle as VO

G = os.path.join(os.path.dirname(__file__), "golden")


def test_vae_param_count_matches_survey():
    n = sum(int(np.prod(s)) for s in VO.vae_param_shapes().values())
    assert abs(n - 126.9e6) < 0.1e6          # 127 M params


def test_vae_decode_encode_match_reference():
    g = dict(np.load(os.path.join(G, "vae_small.npz")))
    W = VO.synth_vae_weights()
    scale = VO.default_scale()
    gen = torch.Generator().manual_seed(21)
    z = torch.randn(1, 16, 3, 8, 8, generator=gen)
    with torch.no_grad():
        dec = VO.vae_decode(z, W, scale)
        assert torch.equal(dec, torch.from_numpy(g["dec"]))
        assert torch.equal(VO.float_to_uint8(dec), torch.from_numpy(g["dec_u8"]))
        vid = (torch.rand(1, 3, 9, 64, 64, generator=gen) * 2 - 1)
        vid[:, :, 1:] *= 0.5
        enc = VO.vae_encode(vid, W, scale)
        assert torch.equal(enc, torch.from_numpy(g["enc"]))


def test_vae_spatial_tiling_matches_reference():
    """vae.py:676-717 (spatial_tiled_decode), :769-839 (tiled decode_to_cpu_uint8), :841-881 (spatial_tiled_encode): 3 x 3
    overlapping tiles of a 128 x 128 clip, tile 64 px."""
    g = dict(np.load(os.path.join(G, "vae_tiled.npz")))
    W = VO.synth_vae_weights()
    scale = VO.default_scale()
    gen = torch.Generator().manual_seed(22)
    z = torch.randn(1, 16, 2, 16, 16, generator=gen)
    vid = (torch.rand(1, 3, 5, 128, 128, generator=gen) * 2 - 1)
    vid[:, :, 1:] *= 0.5
    with torch.no_grad():
        assert torch.equal(VO.vae_tiled_decode(z, W, scale, 64), torch.from_numpy(g["dec"]))
        u8 = VO.vae_tiled_decode_uint8(z, W, scale, 64)
        assert torch.equal(u8, torch.from_numpy(g["dec_u8"]))
        assert torch.equal(u8[:, :, 1:4, :100, :120], torch.from_numpy(g["dec_u8_crop"]))     # frame_start / target_* only crop
        assert torch.equal(VO.vae_tiled_encode(vid, W, scale, 64), torch.from_numpy(g["enc"]))


def endframe_inputs():
    gen = torch.Generator().manual_seed(23)
    z = torch.randn(1, 16, 4, 8, 8, generator=gen)
    vid = (torch.rand(1, 3, 10, 64, 64, generator=gen) * 2 - 1)
    vid[:, :, 1:-1] *= 0.5
    return z, vid


def test_vae_any_end_frame_matches_reference():
    """vae.py:590-606 / :646-650: the end image of a start + end conditioned clip bypasses the causal feature cache.  Also pins the
    identity the product relies on: that chunk is an independent one-frame clip, so any_end_frame = (clip without its last
    frame) followed by (the last frame on its own)."""
    g = dict(np.load(os.path.join(G, "vae_endframe.npz")))
    W = VO.synth_vae_weights()
    scale = VO.default_scale()
    z, vid = endframe_inputs()
    with torch.no_grad():
        dec = VO.vae_decode(z, W, scale, any_end_frame=True)
        assert tuple(dec.shape) == (1, 3, 10, 64, 64) and torch.equal(dec, torch.from_numpy(g["dec"]))
        assert torch.equal(VO.float_to_uint8(dec), torch.from_numpy(g["dec_u8"]))
        enc = VO.vae_encode(vid, W, scale, any_end_frame=True)
        assert tuple(enc.shape) == (1, 16, 4, 8, 8) and torch.equal(enc, torch.from_numpy(g["enc"]))
        assert torch.equal(dec, torch.cat([VO.vae_decode(z[:, :, :-1], W, scale), VO.vae_decode(z[:, :, -1:], W, scale)], 2))
        assert torch.equal(enc, torch.cat([VO.vae_encode(vid[:, :, :9], W, scale), VO.vae_encode(vid[:, :, -1:], W, scale)], 2))


def test_fp16_storage_plan_is_off_by_default_and_stays_next_to_the_pinned_plan():
    """`with VO.fp16_plan()` (rounding where the HIP library stores fp16) is a second, UNPINNED plan used only to attribute
    differing bytes (tests/test_gpu_vae_720p.py): outside the context the pinned fp32 restatement is untouched; inside, the
    uint8 frames move by at most 1 LSB in a few percent of the bytes -- the same size as the HIP library's distance to the golden."""
    W, sc = VO.synth_vae_weights(), VO.default_scale()
    g = torch.Generator().manual_seed(21)
    z = torch.randn(1, 16, 3, 8, 8, generator=g)
    with torch.no_grad():
        a = VO.vae_decode(z, W, sc)
        with VO.fp16_plan():
            b = VO.vae_decode(z, W, sc)
        c = VO.vae_decode(z, W, sc)
    assert torch.equal(a, c) and not torch.equal(a, b)
    d = (VO.float_to_uint8(a).int() - VO.float_to_uint8(b).int()).abs()
    assert int(d.max()) == 1 and 0.85 <= (d == 0).float().mean().item() <= 0.99
