"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Wan2.1 causal 3D VAE.

Restates models/wan/modules/vae.py (file:line citations into /root/reference):
  CausalConv3d :43-82, RMS_norm :85-103, Upsample :105-111, Resample :114-212,
  ResidualBlock :238-273, AttentionBlock :276-315, Encoder3d :318-428, Decoder3d :430-538,
  WanVAE_.encode :586-625, .decode :628-662, _vae_float_to_cpu_uint8 :18-20,
  WanVAE mean/std constants :948-958.
Functional, NCTHW tensors, weights = the reference's state_dict keys.  Pinned bit-exactly (fp32)
against the reference's own module on tests/golden/vae_small.npz (oracle/make_golden_vae.py).
Only used by tests / bench cpu_baseline / smoke -- never by the product package.

Second plan, `with fp16_plan():` -- the SAME graph with every tensor the HIP library stores in fp16 rounded to fp16 at the
point where it is stored (wan2gp_amd/vae.py, csrc/vae_graph.hip: the packed input, every convolution output after bias /
residual, RMS_norm(+SiLU), the attention block's q|k, V^T, scaled scores, probabilities and P.V), arithmetic in fp32 in
between, the decoder head kept in fp32.  It is NOT pinned to the reference (the reference has no such mode on a CPU);
it exists so that a test can attribute the bytes in which the HIP decode differs from the fp32 golden to the fp16
storage plan rather than to a kernel (tests/test_gpu_vae_720p.py).  Outside the context manager nothing changes.
"""
import contextlib
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

CACHE_T = 2
MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
        0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
       3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]

CFG = dict(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, temperal_downsample=[False, True, True])

_PLAN16 = [False]


@contextlib.contextmanager
def fp16_plan():
    """Round every tensor the HIP VAE stores in fp16 to fp16 where it is stored (see the module docstring)."""
    old = _PLAN16[0]
    _PLAN16[0] = True
    try:
        yield
    finally:
        _PLAN16[0] = old


def _q(x):
    return x.to(torch.float16).to(x.dtype) if _PLAN16[0] else x


def _conv_taps(x, w, b, stride):
    """conv3d of an already padded x [B,C,T,H,W] with w [Co,Ci,kt,kh,kw] as the sum over the taps of one fp32 matmul each.
    Used when the tensors live on a GPU (tests/test_gpu_vae_720p.py runs this restatement at 720 x 1280 there: the box's CPU
    needs minutes per pass at that size): same products, fp32 accumulation, only the summation order differs from
    F.conv3d -- no vendor convolution library involved.  Pinned to the CPU execution at the golden size by that test."""
    kt, kh, kw = w.shape[2:]
    st, sh, sw = stride
    To, Ho, Wo = (x.shape[2] - kt) // st + 1, (x.shape[3] - kh) // sh + 1, (x.shape[4] - kw) // sw + 1
    out = None
    for a in range(kt):
        for i in range(kh):
            for j in range(kw):
                xs = x[:, :, a:a + (To - 1) * st + 1:st, i:i + (Ho - 1) * sh + 1:sh, j:j + (Wo - 1) * sw + 1:sw]
                y = torch.einsum("oc,bcthw->bothw", w[:, :, a, i, j], xs)
                out = y if out is None else out + y
    return out if b is None else out + b.view(1, -1, 1, 1, 1)


def _conv3d(x, w, b, stride=(1, 1, 1)):
    return _conv_taps(x, w, b, stride) if x.is_cuda else F.conv3d(x, w, b, stride=stride)


def _conv2d(x, w, b, stride=1, padding=0):
    if not x.is_cuda:
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    if padding:
        x = F.pad(x, (padding,) * 4)
    return _conv_taps(x.unsqueeze(2), w.unsqueeze(2), b, (1, stride, stride)).squeeze(2)


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
    return _conv3d(x, w, b, stride)


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
    h = _q(causal_conv3d(x, W[p + "shortcut.weight"], W[p + "shortcut.bias"])) if (p + "shortcut.weight") in W else x
    for n, c in (("0", "2"), ("3", "6")):
        x = _q(F.silu(rms_norm(x, W[p + f"residual.{n}.gamma"])))
        if cache is not None:
            cx = _cache_update(x, cache[idx[0]])
            x = causal_conv3d(x, W[p + f"residual.{c}.weight"], W[p + f"residual.{c}.bias"], cache[idx[0]])
            cache[idx[0]] = cx
            idx[0] += 1
        else:
            x = causal_conv3d(x, W[p + f"residual.{c}.weight"], W[p + f"residual.{c}.bias"])
        if c == "2":
            x = _q(x)           # fp16 plan: the second convolution's rounding comes after the fused residual add
    return _q(x + h)


def attention_block(x, W, p):
    """AttentionBlock.forward (vae.py:294-315): per-frame single-head attention over h*w tokens."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = _q(rms_norm(y, W[p + "norm.gamma"]))
    qkv = _q(_conv2d(y, W[p + "to_qkv.weight"], W[p + "to_qkv.bias"]))
    q, k, v = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    if _PLAN16[0] or x.is_cuda:  # fp16 plan: scores, probabilities and P.V are fp16 tensors in HBM (wan2gp_amd/vae.py attention_block);
        sc = _q((q @ k.transpose(-1, -2)) * (1.0 / c ** 0.5))   # on a GPU (either plan): plain matmuls instead of a vendor attention kernel
        y = _q(_q(torch.softmax(sc, dim=-1)) @ v)
    else:
        y = F.scaled_dot_product_attention(q, k, v)
    y = y.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    y = _conv2d(y, W[p + "proj.weight"], W[p + "proj.bias"])
    y = y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return _q(y + x)


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
                x = _q(causal_conv3d(x, W[p + "time_conv.weight"], W[p + "time_conv.bias"], pad=(1, 0, 0)))
            else:
                x = _q(causal_conv3d(x, W[p + "time_conv.weight"], W[p + "time_conv.bias"], cache[i], pad=(1, 0, 0)))
            cache[i] = cache_x
            idx[0] += 1
            x = x.reshape(b, 2, c, t, h, w)
            x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, w)
    t = x.shape[2]
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    if mode in ("upsample2d", "upsample3d"):
        y = F.interpolate(y.float(), scale_factor=(2., 2.), mode="nearest-exact").type_as(y)
        y = _q(_conv2d(y, W[p + "resample.1.weight"], W[p + "resample.1.bias"], padding=1))
    else:
        y = _q(_conv2d(F.pad(y, (0, 1, 0, 1)), W[p + "resample.1.weight"], W[p + "resample.1.bias"], stride=2))
    x = y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)
    if mode == "downsample3d" and cache is not None:
        i = idx[0]
        if cache[i] is None:
            cache[i] = x
            idx[0] += 1
        else:
            cache_x = x[:, :, -1:].clone()
            x = _q(causal_conv3d(torch.cat([cache[i][:, :, -1:], x], 2), W[p + "time_conv.weight"], W[p + "time_conv.bias"],
                                 stride=(2, 1, 1), pad=(0, 0, 0)))
            cache[i] = cache_x
            idx[0] += 1
    return x


def _cached_conv(x, W, name, cache, idx):
    q = (lambda v: v) if name == "decoder.head.2" else _q        # the decoder head writes fp32 in the HIP library as well
    if cache is None:
        return q(causal_conv3d(x, W[name + ".weight"], W[name + ".bias"]))
    cx = _cache_update(x, cache[idx[0]])
    y = q(causal_conv3d(x, W[name + ".weight"], W[name + ".bias"], cache[idx[0]]))
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
    x = _q(F.silu(rms_norm(x, W["decoder.head.0.gamma"])))
    return _cached_conv(x, W, "decoder.head.2", cache, idx)


def encoder_forward(x, W, cache, idx, cfg=CFG):
    """Encoder3d.forward (vae.py:371-428)."""
    x = _cached_conv(x, W, "encoder.conv1", cache, idx)
    for kind, p, mode in encoder_layers(cfg):
        x = residual_block(x, W, p, cache, idx) if kind == "res" else resample(x, W, p, mode, cache, idx)
    x = residual_block(x, W, "encoder.middle.0.", cache, idx)
    x = attention_block(x, W, "encoder.middle.1.")
    x = residual_block(x, W, "encoder.middle.2.", cache, idx)
    x = _q(F.silu(rms_norm(x, W["encoder.head.0.gamma"])))
    return _cached_conv(x, W, "encoder.head.2", cache, idx)


def _n_cached_convs(W, side):
    return sum(1 for k, v in W.items() if k.startswith(side) and k.endswith(".weight") and v.dim() == 5)


def vae_decode(z, W, scale=None, cfg=CFG, any_end_frame=False):
    """WanVAE_.decode (vae.py:628-662): one latent frame at a time through the cached decoder.  any_end_frame (:646-650): the
    last latent frame -- the end image of a start + end conditioned clip -- goes through the decoder WITHOUT the feature cache
    (feat_cache=None: causal zero padding, no temporal upsampling), i.e. it decodes to one frame like the first."""
    if scale is not None:
        z = z / scale[1].view(1, -1, 1, 1, 1) + scale[0].view(1, -1, 1, 1, 1)
    x = _q(causal_conv3d(_q(z), W["conv2.weight"], W["conv2.bias"]))
    n = _n_cached_convs(W, "decoder.")
    cache = [None] * n
    outs = []
    it = z.shape[2]
    for i in range(it):
        if any_end_frame and i == it - 1 and i > 0:
            outs.append(decoder_forward(x[:, :, -1:], W, [None] * n, [0], cfg))      # a cache of its own = no cache
        else:
            outs.append(decoder_forward(x[:, :, i:i + 1], W, cache, [0], cfg))
    return torch.cat(outs, 2)


def vae_encode(x, W, scale=None, cfg=CFG, any_end_frame=False):
    """WanVAE_.encode (vae.py:586-625): chunks of 1,4,4,... frames; returns the normalised mu.  any_end_frame (:590-606):
    2 + (t - 2) // 4 chunks, the last one being the clip's last frame alone, encoded without the feature cache."""
    t = x.shape[2]
    x = _q(x)
    n = _n_cached_convs(W, "encoder.")
    cache = [None] * n
    outs = []
    it = 2 + (t - 2) // 4 if any_end_frame else 1 + (t - 1) // 4
    for i in range(it):
        if i == 0:
            outs.append(encoder_forward(x[:, :, :1], W, cache, [0], cfg))
        elif any_end_frame and i == it - 1:
            outs.append(encoder_forward(x[:, :, -1:], W, [None] * n, [0], cfg))
        else:
            outs.append(encoder_forward(x[:, :, 1 + 4 * (i - 1):1 + 4 * i], W, cache, [0], cfg))
    out = torch.cat(outs, 2)
    mu, _ = _q(causal_conv3d(out, W["conv1.weight"], W["conv1.bias"])).chunk(2, dim=1)
    if scale is not None:
        mu = (mu - scale[0].view(1, -1, 1, 1, 1)) * scale[1].view(1, -1, 1, 1, 1)
    return mu


# ---- spatial tiling (vae.py:664-717 decode, :769-839 decode_to_cpu_uint8, :841-881 encode) -------------------------------
def _blend_v(a, b, be):
    """blend_v (vae.py:664-668): rows 0..be-1 of b become a[-be + y] * (1 - y/be) + b[y] * (y/be), in place (Python-float weights)."""
    be = min(a.shape[-2], b.shape[-2], be)
    for y in range(be):
        b[:, :, :, y, :] = a[:, :, :, -be + y, :] * (1 - y / be) + b[:, :, :, y, :] * (y / be)
    return b


def _blend_h(a, b, be):
    be = min(a.shape[-1], b.shape[-1], be)
    for x in range(be):
        b[:, :, :, :, x] = a[:, :, :, :, -be + x] * (1 - x / be) + b[:, :, :, :, x] * (x / be)
    return b


def _blend_tiles(rows, be, row_limit):
    out_rows = []
    for i, row in enumerate(rows):
        out = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend_v(rows[i - 1][j], tile, be)
            if j > 0:
                tile = _blend_h(row[j - 1], tile, be)
            out.append(tile[:, :, :, :row_limit, :row_limit])
        out_rows.append(torch.cat(out, dim=-1))
    return torch.cat(out_rows, dim=-2)


def vae_tiled_decode(z, W, scale, tile_size, cfg=CFG):
    """spatial_tiled_decode (vae.py:676-717): overlapping latent tiles decoded independently, seams blended over tile_size/4 px."""
    tl = int(tile_size / 8)
    z = z / scale[1].view(1, -1, 1, 1, 1) + scale[0].view(1, -1, 1, 1, 1)
    ov = int(tl * 0.75)
    be = int(tile_size * 0.25)
    rows = [[vae_decode(z[:, :, :, i:i + tl, j:j + tl], W, None, cfg) for j in range(0, z.shape[-1], ov)]
            for i in range(0, z.shape[-2], ov)]
    return _blend_tiles(rows, be, tile_size - be)


def _blend_edge(edge, tile, be, dim):
    """_blend_v_edge_ / _blend_h_edge_ (vae.py:23-40): the streaming form of the blend (fp32 tensor weights)."""
    be = min(int(edge.shape[dim]), int(tile.shape[dim]), int(be))
    if be <= 0:
        return
    shape = [1] * 5
    shape[dim] = be
    w = torch.arange(be, dtype=tile.dtype).div_(be).view(shape)
    e = edge.narrow(dim, edge.shape[dim] - be, be).clone()
    e.mul_(1.0 - w)
    tile.narrow(dim, 0, be).mul_(w).add_(e)


def vae_tiled_decode_uint8(z, W, scale, tile_size, cfg=CFG):
    """The tiled branch of decode_to_cpu_uint8 (vae.py:769-839) for the whole clip: tiles in row-major order, each blended with
    the saved bottom edge of the tile above and the right edge of the tile to its left (edges taken AFTER blending), cropped
    to row_limit and converted to uint8."""
    tl = max(1, int(tile_size / 8))
    ov = max(1, int(tl * 0.75))
    be = int(tile_size * 0.25)
    row_limit = max(1, tile_size - be)
    T, H, Wd = (z.shape[2] - 1) * 4 + 1, z.shape[-2] * 8, z.shape[-1] * 8
    out = torch.empty(z.shape[0], 3, T, H, Wd, dtype=torch.uint8)
    prev_edges, r = [], 0
    for ly in range(0, z.shape[-2], ov):
        y0, y1 = r * row_limit, min(r * row_limit + row_limit, H)
        if y1 <= y0:
            break
        cur_edges, left, c = [], None, 0
        for lx in range(0, z.shape[-1], ov):
            x0, x1 = c * row_limit, min(c * row_limit + row_limit, Wd)
            if x1 <= x0:
                break
            tz = z[:, :, :, ly:ly + tl, lx:lx + tl].clone()
            tz.div_(scale[1].view(1, -1, 1, 1, 1)).add_(scale[0].view(1, -1, 1, 1, 1))
            tile = vae_decode(tz, W, None, cfg)
            if r > 0 and c < len(prev_edges) and prev_edges[c] is not None:
                _blend_edge(prev_edges[c], tile, be, 3)
            if left is not None:
                _blend_edge(left, tile, be, 4)
            cur_edges.append(tile[:, :, :, -min(be, tile.shape[-2]):, :].clone() if y1 < H else None)
            left = tile[:, :, :, :, -min(be, tile.shape[-1]):].clone() if x1 < Wd else None
            tile = tile[:, :, :, :y1 - y0, :x1 - x0]
            out[:, :, :, y0:y0 + tile.shape[-2], x0:x0 + tile.shape[-1]] = float_to_uint8(tile)
            c += 1
        prev_edges = cur_edges
        r += 1
    return out


def vae_tiled_encode(x, W, scale, tile_size, cfg=CFG):
    """spatial_tiled_encode (vae.py:841-881)."""
    tl = int(tile_size / 8)
    ov = int(tile_size * 0.75)
    be = int(tl * 0.25)
    rows = [[vae_encode(x[:, :, :, i:i + tile_size, j:j + tile_size], W, None, cfg) for j in range(0, x.shape[-1], ov)]
            for i in range(0, x.shape[-2], ov)]
    mu = _blend_tiles(rows, be, tl - be)
    return (mu - scale[0].view(1, -1, 1, 1, 1)) * scale[1].view(1, -1, 1, 1, 1)


def float_to_uint8(frames):
    """_vae_float_to_cpu_uint8 (vae.py:18-20): clamp -> +1 -> *127.5 -> round-half-even -> uint8."""
    return frames.clone().clamp_(-1.0, 1.0).add_(1.0).mul_(127.5).round_().clamp_(0.0, 255.0).to(torch.uint8)


def default_scale(dtype=torch.float32):
    return [torch.tensor(MEAN, dtype=dtype), 1.0 / torch.tensor(STD, dtype=dtype)]
