"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Wan2.2 VAE (5B ti2v: z = 48, stride (4,16,16)),
SURVEY.md section 8(f) rank 3.

Restates models/wan/modules/vae2_2.py (file:line citations into /root/reference):
  patchify / unpatchify :299-332, AvgDown3D :335-386, DupUp3D :389-431, Down_ResidualBlock :434-471,
  Up_ResidualBlock :474-516, Encoder3d :519-632, Decoder3d :635-742, WanVAE_.encode :802-842, .decode :845-880,
  Wan2_2_VAE mean / std :1158-1262 and its config (z 48, dim 160, dec_dim 256, temperal_downsample [F,T,T]) :1146-1155.
CausalConv3d, RMS_norm, ResidualBlock, AttentionBlock and Resample are the Wan2.1 ones up to the channel counts
(vae2_2.py:18-296 vs vae.py:43-315; the upsample Conv2d keeps `dim` channels, :108-117), so they are imported from
oracle/vae_oracle.py, which is pinned to the reference on its own fixtures.
Pinned bit-exactly (fp32) against the reference's own WanVAE_ on tests/golden/vae22_small.npz
(oracle/make_golden_vae22.py).  Only used by tests -- never by the product package.
"""
from typing import Dict

import torch
import torch.nn.functional as F

from oracle.vae_oracle import _cached_conv, attention_block, causal_conv3d, float_to_uint8, residual_block, resample, rms_norm  # noqa: F401

MEAN = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891, 0.1570, -0.0098,
        0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502, -0.2055, -0.0322, 0.1109, 0.1567,
        -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723, -0.2839, -0.2083, -0.0520, 0.3748, 0.0152, 0.1957,
        0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
STD = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901, 0.6885, 0.6165, 0.8454,
       0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659, 0.7069, 0.5338, 0.4889, 0.4917, 0.4069, 0.4999,
       0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944, 0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600, 0.3943, 0.5537, 0.5444,
       0.4089, 0.7468, 0.7744]

CFG = dict(dim=160, dec_dim=256, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2, temperal_downsample=[False, True, True])
SMALL = dict(dim=32, dec_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, temperal_downsample=[False, True, True])


def vae22_param_shapes(cfg=CFG) -> Dict[str, tuple]:
    """state_dict key -> shape of WanVAE_(dim, dec_dim, z_dim, ...) (vae2_2.py:753-795)."""
    dim, dec, z = cfg["dim"], cfg["dec_dim"], cfg["z_dim"]
    mult, nres, tds = cfg["dim_mult"], cfg["num_res_blocks"], cfg["temperal_downsample"]
    p = {}

    def res(pre, cin, cout):
        p[pre + "residual.0.gamma"] = (cin, 1, 1, 1)
        p[pre + "residual.2.weight"] = (cout, cin, 3, 3, 3); p[pre + "residual.2.bias"] = (cout,)
        p[pre + "residual.3.gamma"] = (cout, 1, 1, 1)
        p[pre + "residual.6.weight"] = (cout, cout, 3, 3, 3); p[pre + "residual.6.bias"] = (cout,)
        if cin != cout:
            p[pre + "shortcut.weight"] = (cout, cin, 1, 1, 1); p[pre + "shortcut.bias"] = (cout,)

    def attn(pre, c):
        p[pre + "norm.gamma"] = (c, 1, 1)
        p[pre + "to_qkv.weight"] = (3 * c, c, 1, 1); p[pre + "to_qkv.bias"] = (3 * c,)
        p[pre + "proj.weight"] = (c, c, 1, 1); p[pre + "proj.bias"] = (c,)

    dims = [dim * u for u in [1] + list(mult)]
    p["encoder.conv1.weight"] = (dims[0], 12, 3, 3, 3); p["encoder.conv1.bias"] = (dims[0],)
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        pre = f"encoder.downsamples.{i}.downsamples."
        for j in range(nres):
            res(f"{pre}{j}.", cin, cout)
            cin = cout
        if i != len(mult) - 1:
            p[f"{pre}{nres}.resample.1.weight"] = (cout, cout, 3, 3); p[f"{pre}{nres}.resample.1.bias"] = (cout,)
            if tds[i]:
                p[f"{pre}{nres}.time_conv.weight"] = (cout, cout, 3, 1, 1); p[f"{pre}{nres}.time_conv.bias"] = (cout,)
    c = dims[-1]
    res("encoder.middle.0.", c, c); attn("encoder.middle.1.", c); res("encoder.middle.2.", c, c)
    p["encoder.head.0.gamma"] = (c, 1, 1, 1)
    p["encoder.head.2.weight"] = (2 * z, c, 3, 3, 3); p["encoder.head.2.bias"] = (2 * z,)
    p["conv1.weight"] = (2 * z, 2 * z, 1, 1, 1); p["conv1.bias"] = (2 * z,)
    p["conv2.weight"] = (z, z, 1, 1, 1); p["conv2.bias"] = (z,)
    dd = [dec * u for u in [mult[-1]] + list(mult[::-1])]
    tus = list(tds[::-1])
    p["decoder.conv1.weight"] = (dd[0], z, 3, 3, 3); p["decoder.conv1.bias"] = (dd[0],)
    res("decoder.middle.0.", dd[0], dd[0]); attn("decoder.middle.1.", dd[0]); res("decoder.middle.2.", dd[0], dd[0])
    for i, (cin, cout) in enumerate(zip(dd[:-1], dd[1:])):
        pre = f"decoder.upsamples.{i}.upsamples."
        for j in range(nres + 1):
            res(f"{pre}{j}.", cin, cout)
            cin = cout
        if i != len(mult) - 1:
            p[f"{pre}{nres + 1}.resample.1.weight"] = (cout, cout, 3, 3); p[f"{pre}{nres + 1}.resample.1.bias"] = (cout,)
            if tus[i]:
                p[f"{pre}{nres + 1}.time_conv.weight"] = (2 * cout, cout, 3, 1, 1); p[f"{pre}{nres + 1}.time_conv.bias"] = (2 * cout,)
    p["decoder.head.0.gamma"] = (dd[-1], 1, 1, 1)
    p["decoder.head.2.weight"] = (12, dd[-1], 3, 3, 3); p["decoder.head.2.bias"] = (12,)
    return p


def synth_vae22_weights(seed=77, dtype=torch.float32, cfg=SMALL) -> Dict[str, torch.Tensor]:
    """Seeded random weights, fp16-representable (the HIP path stores fp16 weights)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in vae22_param_shapes(cfg).items():
        if k.endswith("gamma"):
            w = 1.0 + 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            w = torch.randn(shp, generator=g) * (1.2 / fan_in ** 0.5)
        out[k] = w.to(torch.float16).to(dtype)
    return out


# ---------------------------------------------------------------------------------------------------------------------
def patchify(x, q=2):
    """vae2_2.py:299-315, 5-D: "b c f (h q) (w r) -> b (c r q) f h w"."""
    b, c, f, H, W = x.shape
    h, w = H // q, W // q
    x = x.view(b, c, f, h, q, w, q)                      # b c f h q w r
    return x.permute(0, 1, 6, 4, 2, 3, 5).reshape(b, c * q * q, f, h, w)   # b c r q f h w


def unpatchify(x, q=2):
    """vae2_2.py:318-332: "b (c r q) f h w -> b c f (h q) (w r)"."""
    b, C, f, h, w = x.shape
    c = C // (q * q)
    x = x.view(b, c, q, q, f, h, w)                      # b c r q f h w
    return x.permute(0, 1, 4, 5, 3, 6, 2).reshape(b, c, f, h * q, w * q)   # b c f h q w r


def avg_down3d(x, cout, ft, fs):
    """AvgDown3D.forward (vae2_2.py:354-386): zero-pad time in FRONT to a multiple of ft, fold the (ft, fs, fs) block into
    the channels (channel-major), average groups of cin*ft*fs*fs/cout consecutive folded channels."""
    pad_t = (ft - x.shape[2] % ft) % ft
    x = F.pad(x, (0, 0, 0, 0, pad_t, 0))
    B, C, T, H, W = x.shape
    x = x.view(B, C, T // ft, ft, H // fs, fs, W // fs, fs).permute(0, 1, 3, 5, 7, 2, 4, 6).contiguous()
    x = x.view(B, cout, C * ft * fs * fs // cout, T // ft, H // fs, W // fs)
    return x.mean(dim=2)


def dup_up3d(x, cout, ft, fs, first_chunk=False):
    """DupUp3D.forward (vae2_2.py:409-431)."""
    B, C, T, H, W = x.shape
    x = x.repeat_interleave(cout * ft * fs * fs // C, dim=1)
    x = x.view(B, cout, ft, fs, fs, T, H, W).permute(0, 1, 5, 2, 6, 3, 7, 4).contiguous()
    x = x.view(B, cout, T * ft, H * fs, W * fs)
    return x[:, :, ft - 1:] if first_chunk else x


def down_block(x, W, pre, cin, cout, nres, t_down, down, cache, idx):
    """Down_ResidualBlock.forward (vae2_2.py:466-471)."""
    x0 = x
    for j in range(nres):
        x = residual_block(x, W, f"{pre}{j}.", cache, idx)
    if down:
        x = resample(x, W, f"{pre}{nres}.", "downsample3d" if t_down else "downsample2d", cache, idx)
    return x + avg_down3d(x0, cout, 2 if t_down else 1, 2 if down else 1)


def up_block(x, W, pre, cin, cout, nres, t_up, up, cache, idx, first_chunk):
    """Up_ResidualBlock.forward (vae2_2.py:508-516)."""
    xm = x
    for j in range(nres + 1):
        xm = residual_block(xm, W, f"{pre}{j}.", cache, idx)
    if not up:
        return xm
    xm = resample(xm, W, f"{pre}{nres + 1}.", "upsample3d" if t_up else "upsample2d", cache, idx)
    return xm + dup_up3d(x, cout, 2 if t_up else 1, 2, first_chunk)


def encoder_forward(x, W, cache, idx, cfg=CFG):
    """Encoder3d.forward (vae2_2.py:578-632)."""
    mult, nres, tds = cfg["dim_mult"], cfg["num_res_blocks"], cfg["temperal_downsample"]
    dims = [cfg["dim"] * u for u in [1] + list(mult)]
    x = _cached_conv(x, W, "encoder.conv1", cache, idx)
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        down = i != len(mult) - 1
        x = down_block(x, W, f"encoder.downsamples.{i}.downsamples.", cin, cout, nres, tds[i] if i < len(tds) else False, down,
                       cache, idx)
    x = residual_block(x, W, "encoder.middle.0.", cache, idx)
    x = attention_block(x, W, "encoder.middle.1.")
    x = residual_block(x, W, "encoder.middle.2.", cache, idx)
    x = F.silu(rms_norm(x, W["encoder.head.0.gamma"]))
    return _cached_conv(x, W, "encoder.head.2", cache, idx)


def decoder_forward(x, W, cache, idx, cfg=CFG, first_chunk=False):
    """Decoder3d.forward (vae2_2.py:691-742)."""
    mult, nres = cfg["dim_mult"], cfg["num_res_blocks"]
    tus = list(cfg["temperal_downsample"][::-1])
    dd = [cfg["dec_dim"] * u for u in [mult[-1]] + list(mult[::-1])]
    x = _cached_conv(x, W, "decoder.conv1", cache, idx)
    x = residual_block(x, W, "decoder.middle.0.", cache, idx)
    x = attention_block(x, W, "decoder.middle.1.")
    x = residual_block(x, W, "decoder.middle.2.", cache, idx)
    for i, (cin, cout) in enumerate(zip(dd[:-1], dd[1:])):
        up = i != len(mult) - 1
        x = up_block(x, W, f"decoder.upsamples.{i}.upsamples.", cin, cout, nres, tus[i] if i < len(tus) else False, up, cache, idx,
                     first_chunk)
    x = F.silu(rms_norm(x, W["decoder.head.0.gamma"]))
    return _cached_conv(x, W, "decoder.head.2", cache, idx)


def _n_cached_convs(W, side):
    return sum(1 for k, v in W.items() if k.startswith(side) and k.endswith(".weight") and v.dim() == 5)


def vae22_encode(x, W, scale=None, cfg=CFG):
    """WanVAE_.encode (vae2_2.py:802-842), any_end_frame=False."""
    x = patchify(x, 2)
    t = x.shape[2]
    cache = [None] * _n_cached_convs(W, "encoder.")
    outs = []
    for i in range(1 + (t - 1) // 4):
        chunk = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(encoder_forward(chunk, W, cache, [0], cfg))
    out = torch.cat(outs, 2)
    mu, _ = causal_conv3d(out, W["conv1.weight"], W["conv1.bias"]).chunk(2, dim=1)
    if scale is not None:
        mu = (mu - scale[0].view(1, -1, 1, 1, 1)) * scale[1].view(1, -1, 1, 1, 1)
    return mu


def vae22_decode(z, W, scale=None, cfg=CFG):
    """WanVAE_.decode (vae2_2.py:845-880), any_end_frame=False."""
    if scale is not None:
        z = z / scale[1].view(1, -1, 1, 1, 1) + scale[0].view(1, -1, 1, 1, 1)
    x = causal_conv3d(z, W["conv2.weight"], W["conv2.bias"])
    cache = [None] * _n_cached_convs(W, "decoder.")
    outs = []
    for i in range(z.shape[2]):
        outs.append(decoder_forward(x[:, :, i:i + 1], W, cache, [0], cfg, first_chunk=(i == 0)))
    return unpatchify(torch.cat(outs, 2), 2)


def default_scale(dtype=torch.float32, z_dim=48):
    return [torch.tensor(MEAN[:z_dim], dtype=dtype), 1.0 / torch.tensor(STD[:z_dim], dtype=dtype)]
