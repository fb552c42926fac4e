"""Pins oracle/vae22_oracle.py (CPU restatement of models/wan/modules/vae2_2.py) to tests/golden/vae22_small.npz, produced by
oracle/make_golden_vae22.py from the reference's own WanVAE_ / patchify / AvgDown3D / DupUp3D.  fp32, bit-exact."""
import os

import numpy as np
import torch

from oracle import vae22_oracle as V2

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "vae22_small.npz")))


def t(name):
    return torch.from_numpy(G[name])


def test_patchify_and_shortcuts_reproduce_reference():
    assert torch.equal(V2.patchify(t("patch_in"), 2), t("patch_out"))
    assert torch.equal(V2.unpatchify(t("patch_out"), 2), t("unpatch_out")) and torch.equal(t("unpatch_out"), t("patch_in"))
    x = t("avg_in")
    assert torch.equal(V2.avg_down3d(x, 16, 2, 2), t("avg_t2s2"))
    assert torch.equal(V2.avg_down3d(x, 16, 1, 2), t("avg_t1s2"))
    assert torch.equal(V2.avg_down3d(x[:, :, :1], 16, 2, 2), t("avg_t2s2_odd"))      # one frame: a zero frame is padded in FRONT
    assert torch.equal(V2.avg_down3d(x, 4, 1, 1), t("avg_t1s1"))
    assert torch.equal(V2.dup_up3d(x, 4, 2, 2), t("dup_t2s2"))
    assert torch.equal(V2.dup_up3d(x, 4, 2, 2, True), t("dup_t2s2_first"))
    assert torch.equal(V2.dup_up3d(x, 4, 1, 2), t("dup_t1s2"))


def test_vae22_decode_encode_reproduce_reference():
    cfg = V2.SMALL
    W = V2.synth_vae22_weights(cfg=cfg)
    scale = V2.default_scale(z_dim=cfg["z_dim"])
    g = torch.Generator().manual_seed(int(G["seed"][0]))
    z = torch.randn(1, cfg["z_dim"], 3, 4, 4, generator=g)
    vid = torch.rand(1, 3, 9, 64, 64, generator=g) * 2 - 1
    vid[:, :, 1:] *= 0.5
    with torch.no_grad():
        dec = V2.vae22_decode(z, W, scale, cfg)
        enc = V2.vae22_encode(vid, W, scale, cfg)
    assert dec.shape == (1, 3, 9, 64, 64) and enc.shape == (1, cfg["z_dim"], 3, 4, 4)
    assert np.array_equal(dec.numpy(), G["dec"])
    assert np.array_equal(V2.float_to_uint8(dec).numpy(), G["dec_u8"])
    assert np.array_equal(enc.numpy(), G["enc"])


def test_param_shapes_full_config():
    p = V2.vae22_param_shapes(V2.CFG)
    assert p["encoder.conv1.weight"] == (160, 12, 3, 3, 3) and p["decoder.head.2.weight"] == (12, 256, 3, 3, 3)
    assert p["decoder.conv1.weight"] == (1024, 48, 3, 3, 3) and p["conv1.weight"] == (96, 96, 1, 1, 1)
    assert p["decoder.upsamples.0.upsamples.3.time_conv.weight"] == (2048, 1024, 3, 1, 1)
    assert "decoder.upsamples.2.upsamples.3.time_conv.weight" not in p and "decoder.upsamples.3.upsamples.3.resample.1.weight" not in p
