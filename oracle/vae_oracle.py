"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Wan2.1 causal 3D VAE.

Restates models/wan/modules/vae.py (file:line citations into /root/reference):
  CausalConv3d :43-82, RMS_norm :85-103, Upsample :105-111, Resample :114-212,
  ResidualBlock :238-273, AttentionBlock :276-315, Encoder3d :318-428, Decoder3d :430-538,
  WanVAE_.encode :586-625, .decode :628-662, _vae_float_to_cpu_uint8 :18-20,
  WanVAE mean/std constants :948-958.
Functional, NCTHW tensors, weights = the reference's state_dict keys.  Pinned bit-exactly (fp32)
against the reference's own module on tests/golden/vae_small.npz (oracle/make_golden_vae.py).
Only used by tests / bench cpu_baseline / smoke -- never by the product package.
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

CACHE_T = 2
MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
        0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
       3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]

CFG = dict(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, temperal_downsample=[False, True, True])


def vae_param_shapes(cfg=CFG) -> Dict[str, tuple]:
    """state_dict key -> shape for WanVAE_(dim=96, z=16, mult [1,2,4,4], attn_scales=[]) (vae.py:906-918)."""
    dim, z = cfg["dim"], cfg["z_dim"]
    mult, nres, tds = cfg["dim_mult"], cfg["num_res_blocks"], cfg["temperal_downsample"]
    p = {}

    def res(prefix, cin, cout):
        p[prefix + "residual.0.gamma"] = (cin, 1, 1, 1)
        p[prefix + "residual.2.weight"] = (cout, cin, 3, 3, 3); p[prefix + "residual.2.bias"] = (cout,)
        p[prefix + "residual.3.gamma"] = (cout, 1, 1, 1)
        p[prefix + "residual.6.weight"] = (cout, cout, 3, 3, 3); p[prefix + "residual.6.bias"] = (cout,)
        if cin != cout:
            p[prefix + "shortcut.weight"] = (cout, cin, 1, 1, 1); p[prefix + "shortcut.bias"] = (cout,)

    def attn(prefix, c):
        p[prefix + "norm.gamma"] = (c, 1, 1)
        p[prefix + "to_qkv.weight"] = (3 * c, c, 1, 1); p[prefix + "to_qkv.bias"] = (3 * c,)
        p[prefix + "proj.weight"] = (c, c, 1, 1); p[prefix + "proj.bias"] = (c,)

    # encoder (vae.py:338-369)
    dims = [dim * u for u in [1] + mult]
    p["encoder.conv1.weight"] = (dims[0], 3, 3, 3, 3); p["encoder.conv1.bias"] = (dims[0],)
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(nres):
            res(f"encoder.downsamples.{idx}.", cin, cout); idx += 1
            cin = cout
        if i != len(mult) - 1:
            p[f"encoder.downsamples.{idx}.resample.1.weight"] = (cout, cout, 3, 3)
            p[f"encoder.downsamples.{idx}.resample.1.bias"] = (cout,)
            if tds[i]:
                p[f"encoder.downsamples.{idx}.time_conv.weight"] = (cout, cout, 3, 1, 1)
                p[f"encoder.downsamples.{idx}.time_conv.bias"] = (cout,)
            idx += 1
    c = dims[-1]
    res("encoder.middle.0.", c, c); attn("encoder.middle.1.", c); res("encoder.middle.2.", c, c)
    p["encoder.head.0.gamma"] = (c, 1, 1, 1)
    p["encoder.head.2.weight"] = (2 * z, c, 3, 3, 3); p["encoder.head.2.bias"] = (2 * z,)
    p["conv1.weight"] = (2 * z, 2 * z, 1, 1, 1); p["conv1.bias"] = (2 * z,)
    p["conv2.weight"] = (z, z, 1, 1, 1); p["conv2.bias"] = (z,)
    # decoder (vae.py:449-484)
    ddims = [dim * u for u in [mult[-1]] + mult[::-1]]
    tus = tds[::-1]
    p["decoder.conv1.weight"] = (ddims[0], z, 3, 3, 3); p["decoder.conv1.bias"] = (ddims[0],)
    res("decoder.middle.0.", ddims[0], ddims[0]); attn("decoder.middle.1.", ddims[0]); res("decoder.middle.2.", ddims[0], ddims[0])
    idx = 0
    for i, (cin, cout) in enumerate(zip(ddims[:-1], ddims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(nres + 1):
            res(f"decoder.upsamples.{idx}.", cin, cout); idx += 1
            cin = cout
        if i != len(mult) - 1:
            p[f"decoder.upsamples.{idx}.resample.1.weight"] = (cout // 2, cout, 3, 3)
            p[f"decoder.upsamples.{idx}.resample.1.bias"] = (cout // 2,)
            if tus[i]:
                p[f"decoder.upsamples.{idx}.time_conv.weight"] = (2 * cout, cout, 3, 1, 1)
                p[f"decoder.upsamples.{idx}.time_conv.bias"] = (2 * cout,)
            idx += 1
    p["decoder.head.0.gamma"] = (ddims[-1], 1, 1, 1)
    p["decoder.head.2.weight"] = (3, ddims[-1], 3, 3, 3); p["decoder.head.2.bias"] = (3,)
    return p


def synth_vae_weights(seed: int = 99, dtype=torch.float32, cfg=CFG) -> Dict[str, torch.Tensor]:
    """Seeded synthetic VAE checkpoint: conv weights ~ N(0, 1/fan_in) * 1.2 (keeps activations O(1)
    through 60 layers), biases N(0,.02), gammas 1+N(0,.05); values rounded through fp16 so the fp16
    and fp32 plans share identical weights."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in vae_param_shapes(cfg).items():
        if k.endswith("gamma"):
            w = 1.0 + 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            w = torch.randn(shp, generator=g) * (1.2 / fan_in ** 0.5)
        out[k] = w.to(torch.float16).to(dtype)
    return out


# ----------------------------------------------------------------------------------------------
def causal_conv3d(x, w, b, cache_x=None, stride=(1, 1, 1), pad=None):
    """CausalConv3d.forward (vae.py:54-61): pad (W,W,H,H,2*T,0), cached frames replace part of the
    temporal zero padding."""
    kt, kh, kw = w.shape[2:]
    if pad is None:
        pad = (kt // 2, kh // 2, kw // 2)
    padding = [pad[2], pad[2], pad[1], pad[1], 2 * pad[0], 0]
    if cache_x is not None and padding[4] > 0:
        x = torch.cat([cache_x, x], dim=2)
        padding[4] -= cache_x.shape[2]
    x = F.pad(x, padding)
    return F.conv3d(x, w, b, stride=stride)


def rms_norm(x, gamma, channel_dim=1):
    """RMS_norm.forward (vae.py:97-103): F.normalize(x, dim=1) * sqrt(C) * gamma."""
    return (F.normalize(x, dim=channel_dim) * (x.shape[channel_dim] ** 0.5) * gamma).to(x.dtype)


def _cache_update(x, old):
    """The `cache_x` bookkeeping shared by every cached conv (e.g. vae.py:256-263)."""
    cache_x = x[:, :, -CACHE_T:].clone()
    if cache_x.shape[2] < 2 and old is not None:
        cache_x = torch.cat([old[:, :, -1:].to(cache_x.device), cache_x], dim=2)
    return cache_x


def residual_block(x, W, p, cache, idx):
    """ResidualBlock.forward (vae.py:251-273)."""
    h = causal_conv3d(x, W[p + "shortcut.weight"], W[p + "shortcut.bias"]) if (p + "shortcut.weight") in W else x
    for n, c in (("0", "2"), ("3", "6")):
        x = F.silu(rms_norm(x, W[p + f"residual.{n}.gamma"]))
        if cache is not None:
            cx = _cache_update(x, cache[idx[0]])
            x = causal_conv3d(x, W[p + f"residual.{c}.weight"], W[p + f"residual.{c}.bias"], cache[idx[0]])
            cache[idx[0]] = cx
            idx[0] += 1
        else:
            x = causal_conv3d(x, W[p + f"residual.{c}.weight"], W[p + f"residual.{c}.bias"])
    return x + h


def attention_block(x, W, p):
    """AttentionBlock.forward (vae.py:294-315): per-frame single-head attention over h*w tokens."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm(y, W[p + "norm.gamma"])
    qkv = F.conv2d(y, W[p + "to_qkv.weight"], W[p + "to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    y = F.scaled_dot_product_attention(q, k, v)
    y = y.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    y = F.conv2d(y, W[p + "proj.weight"], W[p + "proj.bias"])
    y = y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return y + x


def resample(x, W, p, mode, cache, idx):
    """Resample.forward (vae.py:149-212)."""
    b, c, t, h, w = x.shape
    if mode == "upsample3d" and cache is not None:
        i = idx[0]
        if cache[i] is None:
            cache[i] = "Rep"
            idx[0] += 1
        else:
            cache_x = x[:, :, -CACHE_T:]
            if cache_x.shape[2] < 2 and not isinstance(cache[i], str):
                cache_x = torch.cat([cache[i][:, :, -1:], cache_x], dim=2)
            elif cache_x.shape[2] < 2 and isinstance(cache[i], str):
                cache_x = torch.cat([torch.zeros_like(cache_x), cache_x], dim=2)
            else:
                cache_x = cache_x.clone()
            if isinstance(cache[i], str):
                x = causal_conv3d(x, W[p + "time_conv.weight"], W[p + "time_conv.bias"], pad=(1, 0, 0))
            else:
                x = causal_conv3d(x, W[p + "time_conv.weight"], W[p + "time_conv.bias"], cache[i], pad=(1, 0, 0))
            cache[i] = cache_x
            idx[0] += 1
            x = x.reshape(b, 2, c, t, h, w)
            x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, w)
    t = x.shape[2]
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    if mode in ("upsample2d", "upsample3d"):
        y = F.interpolate(y.float(), scale_factor=(2., 2.), mode="nearest-exact").type_as(y)
        y = F.conv2d(y, W[p + "resample.1.weight"], W[p + "resample.1.bias"], padding=1)
    else:
        y = F.conv2d(F.pad(y, (0, 1, 0, 1)), W[p + "resample.1.weight"], W[p + "resample.1.bias"], stride=2)
    x = y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)
    if mode == "downsample3d" and cache is not None:
        i = idx[0]
        if cache[i] is None:
            cache[i] = x
            idx[0] += 1
        else:
            cache_x = x[:, :, -1:].clone()
            x = causal_conv3d(torch.cat([cache[i][:, :, -1:], x], 2), W[p + "time_conv.weight"], W[p + "time_conv.bias"],
                              stride=(2, 1, 1), pad=(0, 0, 0))
            cache[i] = cache_x
            idx[0] += 1
    return x


def _cached_conv(x, W, name, cache, idx):
    if cache is None:
        return causal_conv3d(x, W[name + ".weight"], W[name + ".bias"])
    cx = _cache_update(x, cache[idx[0]])
    y = causal_conv3d(x, W[name + ".weight"], W[name + ".bias"], cache[idx[0]])
    cache[idx[0]] = cx
    idx[0] += 1
    return y


def decoder_layers(cfg=CFG):
    """(kind, prefix, mode) list of Decoder3d.upsamples (vae.py:461-479)."""
    mult, nres = cfg["dim_mult"], cfg["num_res_blocks"]
    tus = cfg["temperal_downsample"][::-1]
    out, idx = [], 0
    for i in range(len(mult)):
        for _ in range(nres + 1):
            out.append(("res", f"decoder.upsamples.{idx}.", None)); idx += 1
        if i != len(mult) - 1:
            out.append(("resample", f"decoder.upsamples.{idx}.", "upsample3d" if tus[i] else "upsample2d")); idx += 1
    return out


def encoder_layers(cfg=CFG):
    mult, nres, tds = cfg["dim_mult"], cfg["num_res_blocks"], cfg["temperal_downsample"]
    out, idx = [], 0
    for i in range(len(mult)):
        for _ in range(nres):
            out.append(("res", f"encoder.downsamples.{idx}.", None)); idx += 1
        if i != len(mult) - 1:
            out.append(("resample", f"encoder.downsamples.{idx}.", "downsample3d" if tds[i] else "downsample2d")); idx += 1
    return out


def decoder_forward(x, W, cache, idx, cfg=CFG):
    """Decoder3d.forward (vae.py:486-538)."""
    x = _cached_conv(x, W, "decoder.conv1", cache, idx)
    x = residual_block(x, W, "decoder.middle.0.", cache, idx)
    x = attention_block(x, W, "decoder.middle.1.")
    x = residual_block(x, W, "decoder.middle.2.", cache, idx)
    for kind, p, mode in decoder_layers(cfg):
        x = residual_block(x, W, p, cache, idx) if kind == "res" else resample(x, W, p, mode, cache, idx)
    x = F.silu(rms_norm(x, W["decoder.head.0.gamma"]))
    return _cached_conv(x, W, "decoder.head.2", cache, idx)


def encoder_forward(x, W, cache, idx, cfg=CFG):
    """Encoder3d.forward (vae.py:371-428)."""
    x = _cached_conv(x, W, "encoder.conv1", cache, idx)
    for kind, p, mode in encoder_layers(cfg):
        x = residual_block(x, W, p, cache, idx) if kind == "res" else resample(x, W, p, mode, cache, idx)
    x = residual_block(x, W, "encoder.middle.0.", cache, idx)
    x = attention_block(x, W, "encoder.middle.1.")
    x = residual_block(x, W, "encoder.middle.2.", cache, idx)
    x = F.silu(rms_norm(x, W["encoder.head.0.gamma"]))
    return _cached_conv(x, W, "encoder.head.2", cache, idx)


def _n_cached_convs(W, side):
    return sum(1 for k, v in W.items() if k.startswith(side) and k.endswith(".weight") and v.dim() == 5)


def vae_decode(z, W, scale=None, cfg=CFG):
    """WanVAE_.decode (vae.py:628-662): one latent frame at a time through the cached decoder."""
    if scale is not None:
        z = z / scale[1].view(1, -1, 1, 1, 1) + scale[0].view(1, -1, 1, 1, 1)
    x = causal_conv3d(z, W["conv2.weight"], W["conv2.bias"])
    cache = [None] * _n_cached_convs(W, "decoder.")
    outs = []
    for i in range(z.shape[2]):
        outs.append(decoder_forward(x[:, :, i:i + 1], W, cache, [0], cfg))
    return torch.cat(outs, 2)


def vae_encode(x, W, scale=None, cfg=CFG):
    """WanVAE_.encode (vae.py:586-625): chunks of 1,4,4,... frames; returns the normalised mu."""
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


def float_to_uint8(frames):
    """_vae_float_to_cpu_uint8 (vae.py:18-20): clamp -> +1 -> *127.5 -> round-half-even -> uint8."""
    return frames.clone().clamp_(-1.0, 1.0).add_(1.0).mul_(127.5).round_().clamp_(0.0, 255.0).to(torch.uint8)


def default_scale(dtype=torch.float32):
    return [torch.tensor(MEAN, dtype=dtype), 1.0 / torch.tensor(STD, dtype=dtype)]
