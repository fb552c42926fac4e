"""WanVAEHIP -- drop-in for the reference `WanVAE` (models/wan/modules/vae.py:935-1027).

Same public surface: `encode(videos, tile_size, any_end_frame)`, `decode(zs, tile_size)`,
`decode_to_cpu_uint8(zs, tile_size, ...)`, `.model.z_dim`, `.device`, `.dtype`, `.scale`,
`get_VAE_tile_size`; lists in, lists out; latents normalised with the mean/std constants of
vae.py:948-958.  The layer graph and the causal feature-cache bookkeeping of Encoder3d /
Decoder3d / WanVAE_.encode / .decode (vae.py:318-662) run here on the host; every tensor op is a
libwanhip kernel on fp16 channels-last activations [T,H,W,C] (vae_ops.hip): implicit-GEMM MFMA
convolutions with the 2-frame cache, fused nearest-2x upsample, fused time interleave, fused
residual add, RMS_norm+SiLU, the attention block as fp16 GEMMs + a row softmax, and the
float->uint8 conversion.  Spatial tiling (tile_size > 0: vae.py:676-717, :769-839, :841-881) exists in the
reference to fit small VRAM; `get_VAE_tile_size` makes the reference's choice (0 on a 288-GB device unless the caller forces a
preset or the picture exceeds 1920 x 1088), and a tile size gets the reference's tiling: overlapping tiles decoded / encoded
independently -- on several GPUs spread over the ranks of `self.sp` -- seams blended over a quarter tile.  `any_end_frame`
(start + end image clips) is the clip without its last frame followed by that frame alone.
"""
import math
from typing import Dict, List, Optional

import torch

from . import lib as _L
from .lib import check, ptr, stream_ptr

MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
        0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
       3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]
F16 = torch.float16


def _pad32(c):
    return (c + 31) // 32 * 32


class _Conv:
    """A packed convolution: weights [Cout_p][Kp] fp16 with K = ((kt*KH+kh)*KW+kw)*Cin_p + c."""

    def __init__(self, w, b, dev, cout_pad=None, dtype=None):
        if w.dim() == 4:                       # Conv2d -> KT = 1
            w = w.unsqueeze(2)
        cout, cin, kt, kh, kw = w.shape
        cin_p = _pad32(cin)
        cout_p = cout_pad or cout
        wp = torch.zeros(cout_p, kt, kh, kw, cin_p, dtype=torch.float32)
        wp[:cout, ..., :cin] = w.detach().float().permute(0, 2, 3, 4, 1)
        K = kt * kh * kw * cin_p
        Kp = (K + 63) // 64 * 64
        flat = torch.zeros(cout_p, Kp, dtype=torch.float32)
        flat[:, :K] = wp.reshape(cout_p, K)
        bb = torch.zeros(cout_p, dtype=torch.float32)
        if b is not None:
            bb[:cout] = b.detach().float()
        self.w = flat.to(device=dev, dtype=dtype or F16).contiguous()
        self.b = bb.to(device=dev, dtype=dtype or F16).contiguous()
        self.cin, self.cout, self.k = cin_p, cout_p, (kt, kh, kw)


class _VaeNet:
    """Holds packed weights + the op helpers; one instance per WanVAEHIP."""

    def __init__(self, sd: Dict[str, torch.Tensor], dev):
        self.dev = dev
        self.lib = _L.load()
        self.convs: Dict[str, _Conv] = {}
        self.gamma: Dict[str, torch.Tensor] = {}
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() in (4, 5):
                name = k[: -len(".weight")]
                pad = _pad32(v.shape[0]) if name == "conv2" else None    # latent 16 -> 32 (48 -> 64) channels for decoder.conv1
                self.convs[name] = _Conv(v, sd.get(name + ".bias"), dev, cout_pad=pad)
            elif k.endswith("gamma"):
                self.gamma[k] = v.detach().reshape(-1).to(device=dev, dtype=F16).contiguous()
        # attention blocks: keep the raw [3C, C] / [C, C] matrices for the GEMM path
        self.attn = {}
        for side in ("encoder.middle.1.", "decoder.middle.1."):
            if side + "to_qkv.weight" in sd:
                C = sd[side + "proj.weight"].shape[0]
                self.attn[side] = dict(
                    C=C,
                    wqkv=sd[side + "to_qkv.weight"].detach().reshape(3 * C, C).to(device=dev, dtype=F16).contiguous(),
                    bqkv=sd[side + "to_qkv.bias"].detach().to(device=dev, dtype=F16).contiguous())

    # ---- ops ---------------------------------------------------------------------------------------
    def conv(self, x, name, cache=None, res=None, out_f32=False, ups=False, interleave=False, st_t=1, st_s=1,
             front=None, pad_s=None):
        c = self.convs[name]
        T, H, W, C = x.shape
        assert C == c.cin, (name, C, c.cin)
        kt, kh, kw = c.k
        if front is None:
            front = kt - 1                                   # causal: 2*padding[0] frames in front (vae.py:49-51)
        if pad_s is None:
            pad_s = kh // 2
        Hin_eff, Win_eff = (2 * H, 2 * W) if ups else (H, W)
        if st_s == 2:                                        # ZeroPad2d((0,1,0,1)) + stride 2 (vae.py:137-139)
            Ho, Wo = (Hin_eff + 1 - kh) // 2 + 1, (Win_eff + 1 - kw) // 2 + 1
        else:
            Ho, Wo = Hin_eff, Win_eff
        To = (T + front - kt) // st_t + 1
        cout = c.cout
        if interleave:
            out = torch.empty(2 * To, Ho, Wo, cout // 2, dtype=F16, device=self.dev)
        elif out_f32:
            out = torch.empty(To, Ho, Wo, cout, dtype=torch.float32, device=self.dev)
        else:
            out = torch.empty(To, Ho, Wo, cout, dtype=F16, device=self.dev)
        check(self.lib.wan_vae_conv3d(ptr(x), ptr(cache), ptr(c.w), ptr(c.b), ptr(res), None if out_f32 else ptr(out),
                                      ptr(out) if out_f32 else None, T, H, W, C, To, Ho, Wo, cout, kt, kh, kw, st_t, st_s,
                                      front, pad_s, 1 if ups else 0, 1 if interleave else 0, stream_ptr()),
              f"wan_vae_conv3d({name})")
        return out

    def norm(self, x, gname, silu=True):
        out = torch.empty_like(x)
        C = x.shape[-1]
        check(self.lib.wan_vae_rmsnorm_silu(ptr(x), ptr(out), ptr(self.gamma[gname]), x.numel() // C, C, 1 if silu else 0,
                                            stream_ptr()), "wan_vae_rmsnorm_silu")
        return out

    def attention_block(self, x, p):
        """AttentionBlock.forward (vae.py:294-315) per frame: tokens = h*w, one head of C channels."""
        a = self.attn[p]
        C = a["C"]
        T, H, W, _ = x.shape
        L = H * W
        Lp = (L + 63) // 64 * 64
        xn = self.norm(x, p + "norm.gamma", silu=False)
        out = torch.empty_like(x)
        lib = self.lib
        if L % 16 != 0:
            raise _L.WanHipError("VAE attention needs h*w % 16 == 0 at the lowest resolution")
        scale = 1.0 / math.sqrt(C)
        wv, bv = a["wqkv"][2 * C:], a["bqkv"][2 * C:]
        for t in range(T):
            xt = xn[t].reshape(L, C)
            qk = torch.empty(L, 2 * C, dtype=F16, device=self.dev)               # [q | k] = x Wqk^T + b
            check(lib.wan_gemm_f16(ptr(xt), C, ptr(a["wqkv"]), C, ptr(a["bqkv"]), ptr(qk), 2 * C, L, 2 * C, C, 1.0, 0,
                                   stream_ptr()), "vae qk")
            vt = torch.zeros(C, Lp, dtype=F16, device=self.dev)                   # V^T, zero padded columns
            check(lib.wan_gemm_f16(ptr(xt), C, ptr(wv), C, ptr(bv), ptr(vt), Lp, L, C, C, 1.0, 1, stream_ptr()), "vae v^T")
            S = torch.empty(L, Lp, dtype=F16, device=self.dev)                    # q k^T / sqrt(C)
            check(lib.wan_gemm_f16(ptr(qk), 2 * C, ptr(qk[:, C:]), 2 * C, None, ptr(S), Lp, L, L, C, scale, 0,
                                   stream_ptr()), "vae qk^T")
            check(lib.wan_vae_softmax(ptr(S), ptr(S), L, L, Lp, stream_ptr()), "vae softmax")
            o = torch.empty(L, C, dtype=F16, device=self.dev)
            check(lib.wan_gemm_f16(ptr(S), Lp, ptr(vt), Lp, None, ptr(o), C, L, C, Lp, 1.0, 0, stream_ptr()), "vae pv")
            out[t] = self.conv(o.view(1, H, W, C), p + "proj", res=x[t:t + 1].contiguous())[0]
        return out


class _VaeNetF32(_VaeNet):
    """The fp32 plan (`vae_precision` "32", wgp.py:4038 -> WanVAE(dtype=torch.float32), models/wan/modules/vae.py): the same layer
    graph on fp32 channels-last activations and fp32 weights through the fp32 ops of csrc/vae_f32.hip -- no 16-bit rounding point
    anywhere.  Plain FMA kernels (a 720p x 81-frame decode is a minute-class job): an option, not the default."""

    def __init__(self, sd: Dict[str, torch.Tensor], dev):
        self.dev = dev
        self.lib = _L.load()
        self.convs: Dict[str, _Conv] = {}
        self.gamma: Dict[str, torch.Tensor] = {}
        f32 = torch.float32
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() in (4, 5):
                name = k[: -len(".weight")]
                pad = _pad32(v.shape[0]) if name == "conv2" else None
                c = _Conv(v, sd.get(name + ".bias"), "cpu", cout_pad=pad, dtype=f32)
                c.w, c.b = c.w.to(dev), c.b.to(dev)
                self.convs[name] = c
            elif k.endswith("gamma"):
                self.gamma[k] = v.detach().reshape(-1).to(device=dev, dtype=f32).contiguous()
        self.attn = {}
        for side in ("encoder.middle.1.", "decoder.middle.1."):
            if side + "to_qkv.weight" in sd:
                C = sd[side + "proj.weight"].shape[0]
                self.attn[side] = dict(C=C, wqkv=sd[side + "to_qkv.weight"].detach().reshape(3 * C, C).to(device=dev, dtype=f32).contiguous(),
                                       bqkv=sd[side + "to_qkv.bias"].detach().to(device=dev, dtype=f32).contiguous())

    def conv(self, x, name, cache=None, res=None, out_f32=False, ups=False, interleave=False, st_t=1, st_s=1, front=None, pad_s=None):
        c = self.convs[name]
        T, H, W, C = x.shape
        assert C == c.cin and x.dtype == torch.float32, (name, C, c.cin, x.dtype)
        kt, kh, kw = c.k
        front = kt - 1 if front is None else front
        pad_s = kh // 2 if pad_s is None else pad_s
        He, We = (2 * H, 2 * W) if ups else (H, W)
        Ho, Wo = ((He + 1 - kh) // 2 + 1, (We + 1 - kw) // 2 + 1) if st_s == 2 else (He, We)
        To = (T + front - kt) // st_t + 1
        out = torch.empty((2 * To, Ho, Wo, c.cout // 2) if interleave else (To, Ho, Wo, c.cout), dtype=torch.float32, device=self.dev)
        check(self.lib.wan_vae_conv3d_f32(ptr(x), ptr(cache), ptr(c.w), c.w.shape[1], ptr(c.b), ptr(res), ptr(out), T, H, W, C, To, Ho, Wo, c.cout,
                                          kt, kh, kw, st_t, st_s, front, pad_s, 1 if ups else 0, 1 if interleave else 0, stream_ptr()),
              f"wan_vae_conv3d_f32({name})")
        return out

    def norm(self, x, gname, silu=True):
        out = torch.empty_like(x)
        C = x.shape[-1]
        check(self.lib.wan_vae_rmsnorm_silu_f32(ptr(x), ptr(out), ptr(self.gamma[gname]), x.numel() // C, C, 1 if silu else 0, stream_ptr()),
              "wan_vae_rmsnorm_silu_f32")
        return out

    def attention_block(self, x, p):
        """AttentionBlock.forward (vae.py:294-315) per frame in fp32: [q | k | v] = x W^T + b, softmax(q k^T / sqrt(C)) v, proj + x."""
        a = self.attn[p]
        C = a["C"]
        T, H, W, _ = x.shape
        L = H * W
        xn = self.norm(x, p + "norm.gamma", silu=False)
        out = torch.empty_like(x)
        lib, f32 = self.lib, torch.float32
        for t in range(T):
            xt = xn[t].reshape(L, C)
            qkv = torch.empty(L, 3 * C, dtype=f32, device=self.dev)
            check(lib.wan_gemm_f32(ptr(xt), C, ptr(a["wqkv"]), C, 1, ptr(a["bqkv"]), ptr(qkv), 3 * C, L, 3 * C, C, 1.0, stream_ptr()), "vae qkv (fp32)")
            S = torch.empty(L, L, dtype=f32, device=self.dev)
            check(lib.wan_gemm_f32(ptr(qkv), 3 * C, ptr(qkv[:, C:]), 3 * C, 1, None, ptr(S), L, L, L, C, 1.0 / math.sqrt(C), stream_ptr()), "vae q k^T (fp32)")
            check(lib.wan_vae_softmax_f32(ptr(S), L, L, L, stream_ptr()), "vae softmax (fp32)")
            o = torch.empty(L, C, dtype=f32, device=self.dev)
            check(lib.wan_gemm_f32(ptr(S), L, ptr(qkv[:, 2 * C:]), 3 * C, 0, None, ptr(o), C, L, C, L, 1.0, stream_ptr()), "vae p v (fp32)")
            out[t] = self.conv(o.view(1, H, W, C), p + "proj", res=x[t:t + 1].contiguous())[0]
        return out


def vae_param_shapes(dim=96, z=16, mult=(1, 2, 4, 4), nres=2, tds=(False, True, True)):
    """state_dict key -> shape of the Wan2.1 VAE (WanVAE_(dim=96, z_dim=16, ...), vae.py:906-918); used to
    build random-init weights for benchmarking and to validate checkpoints."""
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

    dims = [dim * u for u in (1,) + tuple(mult)]
    p["encoder.conv1.weight"] = (dims[0], 3, 3, 3, 3); p["encoder.conv1.bias"] = (dims[0],)
    li = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(nres):
            res(f"encoder.downsamples.{li}.", cin, cout); li += 1
            cin = cout
        if i != len(mult) - 1:
            p[f"encoder.downsamples.{li}.resample.1.weight"] = (cout, cout, 3, 3); p[f"encoder.downsamples.{li}.resample.1.bias"] = (cout,)
            if tds[i]:
                p[f"encoder.downsamples.{li}.time_conv.weight"] = (cout, cout, 3, 1, 1); p[f"encoder.downsamples.{li}.time_conv.bias"] = (cout,)
            li += 1
    c = dims[-1]
    res("encoder.middle.0.", c, c); attn("encoder.middle.1.", c); res("encoder.middle.2.", c, c)
    p["encoder.head.0.gamma"] = (c, 1, 1, 1)
    p["encoder.head.2.weight"] = (2 * z, c, 3, 3, 3); p["encoder.head.2.bias"] = (2 * z,)
    p["conv1.weight"] = (2 * z, 2 * z, 1, 1, 1); p["conv1.bias"] = (2 * z,)
    p["conv2.weight"] = (z, z, 1, 1, 1); p["conv2.bias"] = (z,)
    dd = [dim * u for u in (mult[-1],) + tuple(mult[::-1])]
    tus = tuple(tds[::-1])
    p["decoder.conv1.weight"] = (dd[0], z, 3, 3, 3); p["decoder.conv1.bias"] = (dd[0],)
    res("decoder.middle.0.", dd[0], dd[0]); attn("decoder.middle.1.", dd[0]); res("decoder.middle.2.", dd[0], dd[0])
    li = 0
    for i, (cin, cout) in enumerate(zip(dd[:-1], dd[1:])):
        if i in (1, 2, 3):
            cin //= 2
        for _ in range(nres + 1):
            res(f"decoder.upsamples.{li}.", cin, cout); li += 1
            cin = cout
        if i != len(mult) - 1:
            p[f"decoder.upsamples.{li}.resample.1.weight"] = (cout // 2, cout, 3, 3); p[f"decoder.upsamples.{li}.resample.1.bias"] = (cout // 2,)
            if tus[i]:
                p[f"decoder.upsamples.{li}.time_conv.weight"] = (2 * cout, cout, 3, 1, 1); p[f"decoder.upsamples.{li}.time_conv.bias"] = (2 * cout,)
            li += 1
    p["decoder.head.0.gamma"] = (dd[-1], 1, 1, 1)
    p["decoder.head.2.weight"] = (3, dd[-1], 3, 3, 3); p["decoder.head.2.bias"] = (3,)
    return p


def random_vae_state_dict(seed=0):
    """Random-init VAE weights of the reference architecture (benchmarking without checkpoints)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in vae_param_shapes().items():
        if k.endswith("gamma"):
            sd[k] = 1.0 + 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.02 * torch.randn(shp, generator=g)
        else:
            fan = 1
            for d in shp[1:]:
                fan *= d
            sd[k] = torch.randn(shp, generator=g) * (1.2 / fan ** 0.5)
    return sd


class _NativeGraph:
    """`wan_vae_*` of the C ABI: the whole encode / decode as one library call each (csrc/vae_graph.hip runs the layer graph and the
    cache bookkeeping on the op-level entry points `_VaeNet` uses).  Weights are handed over as host fp32 arrays and packed inside
    the library; the workspace is a torch-owned byte tensor sized by `wan_vae_workspace_bytes`."""

    def __init__(self, sd, dev):
        from ctypes import byref, c_void_p
        self.lib, self.dev, self._ws = _L.load(), dev, None
        h = c_void_p()
        check(self.lib.wan_vae_create(byref(h)), "wan_vae_create")
        self._h = h
        f32 = lambda t: t.detach().to(device="cpu", dtype=torch.float32).contiguous()
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() in (4, 5):
                name = k[: -len(".weight")]
                w = f32(v if v.dim() == 5 else v.unsqueeze(2))
                b = sd.get(name + ".bias")
                bb = None if b is None else f32(b)
                cout, cin, kt, kh, kw = w.shape
                check(self.lib.wan_vae_set_conv(h, name.encode(), ptr(w), cout, cin, kt, kh, kw, ptr(bb), _pad32(cout) if name == "conv2" else 0),
                      f"wan_vae_set_conv({name})")
            elif k.endswith("gamma"):
                g = f32(v).reshape(-1)
                check(self.lib.wan_vae_set_gamma(h, k.encode(), ptr(g), g.numel()), f"wan_vae_set_gamma({k})")
        for side in ("encoder.middle.1.", "decoder.middle.1."):
            if side + "to_qkv.weight" in sd:
                C = sd[side + "proj.weight"].shape[0]
                wq, bq = f32(sd[side + "to_qkv.weight"]).reshape(3 * C, C).contiguous(), f32(sd[side + "to_qkv.bias"])
                check(self.lib.wan_vae_set_attention(h, side.encode(), ptr(wq), ptr(bq), C), "wan_vae_set_attention")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.lib.wan_vae_destroy(h)

    def _workspace(self, decode, t, h, w):
        need = self.lib.wan_vae_workspace_bytes(self._h, 1 if decode else 0, t, h, w)
        if need < 0:
            raise _L.WanHipError("wan_vae_workspace_bytes: " + self.lib.wan_last_error().decode("utf-8", "replace"))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        return self._ws

    def decode(self, z, want_u8, want_f32):
        C, t, h, w = z.shape
        ws = self._workspace(True, t, h, w)
        T, H, W = (t - 1) * 4 + 1, h * 8, w * 8
        u8 = torch.empty(3, T, H, W, dtype=torch.uint8, device=self.dev) if want_u8 else None
        f32 = torch.empty(3, T, H, W, dtype=torch.float32, device=self.dev) if want_f32 else None
        check(self.lib.wan_vae_decode(self._h, ptr(z), t, h, w, ptr(u8), ptr(f32), ptr(ws), ws.numel(), stream_ptr()), "wan_vae_decode")
        return u8, f32

    def encode(self, v):
        C, T, H, W = v.shape
        ws = self._workspace(False, T, H, W)
        out = torch.empty(16, 1 + (T - 1) // 4, H // 8, W // 8, dtype=torch.float32, device=self.dev)
        check(self.lib.wan_vae_encode(self._h, ptr(v), T, H, W, ptr(out), ptr(ws), ws.numel(), stream_ptr()), "wan_vae_encode")
        return out


def _cache_update(x, old):
    """cache_x bookkeeping (vae.py:256-263): last 2 frames of [old ; x]."""
    if x.shape[0] >= 2:
        return x[-2:].clone()
    if old is None:
        return torch.cat([torch.zeros_like(x[-1:]), x[-1:]], 0)
    return torch.cat([old[-1:], x[-1:]], 0)


class WanVAEHIP:
    CFG = dict(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, temperal_downsample=[False, True, True])

    def __init__(self, z_dim=16, vae_pth=None, dtype=torch.float16, device="cuda", state_dict=None, **unused):
        self.dtype, self.device, self.z_dim = dtype, torch.device(device), z_dim
        self.mean = torch.tensor(MEAN, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(STD, dtype=torch.float32, device=self.device)
        self.scale = [self.mean, 1.0 / self.std]
        self.model = self                                  # reference code reads vae.model.z_dim
        self.upsampler_factor = 1
        self.net = None
        # multi-GPU: a process group view (rank, world, group -- e.g. wan2gp_amd.sp.SequenceParallel) over which the spatial tiles
        # of a tiled decode / encode are spread; None = every tile on this device
        self.sp = None
        if state_dict is not None:
            self.load_state_dict(state_dict)
        elif vae_pth is not None:
            from safetensors.torch import load_file
            self.load_state_dict(load_file(vae_pth))

    NATIVE_GRAPH = True                                    # run encode / decode through wan_vae_* (the Wan2.2 subclass keeps the host graph)
    SUPPORTS_F32 = True                                    # a subclass with a graph of its own must bring its own fp32 pieces (vae22.py does) or say False

    def load_state_dict(self, sd):
        if self.dtype == torch.float32:                    # `vae_precision` "32" (wgp.py:4038): the fp32 plan, on the host graph below
            if not self.SUPPORTS_F32:                      # (round-4 advisor: a subclass silently got the Wan2.1 fp32 graph)
                raise NotImplementedError(f"{type(self).__name__}: the fp32 plan (vae_precision '32') is not implemented for this VAE; "
                                          "construct it with dtype=torch.float16")
            self.net = _VaeNetF32(sd, self.device)
            self.native = None
            return self
        self.net = _VaeNet(sd, self.device)
        self.native = _NativeGraph(sd, self.device) if self.NATIVE_GRAPH else None
        return self

    @property
    def _adt(self):
        """The activation dtype of the plan in use."""
        return torch.float32 if self.dtype == torch.float32 else F16

    def _pack(self, src, out, mul, add, C, Cp, n):
        fn = self.net.lib.wan_vae_pack_f32 if self.dtype == torch.float32 else self.net.lib.wan_vae_pack
        check(fn(ptr(src), ptr(out), ptr(mul), ptr(add), C, Cp, n, stream_ptr()), "wan_vae_pack")

    def _unpack(self, src, out, sub, mul, C, Cs, n):
        fn = self.net.lib.wan_vae_unpack_f32 if self.dtype == torch.float32 else self.net.lib.wan_vae_unpack
        check(fn(ptr(src), ptr(out), ptr(sub), ptr(mul), C, Cs, n, stream_ptr()), "wan_vae_unpack")

    @staticmethod
    def get_VAE_tile_size(vae_config, device_mem_capacity, mixed_precision, output_height=None, output_width=None):
        """vae.py:969-1001.  Automatic (vae_config 0): no tiling from 24 GB up -- 1024-px tiles beyond 1920 x 1088 pixels -- and
        512 / 256 / 128 below 24 / 16 / 8 GB; a preset the user forced (1, 2, 3) keeps its tile size (512 / 256 / 128) even on a
        288-GB device: tiled and untiled decodes are different pictures, so the choice is the caller's, not the backend's."""
        if vae_config == 0:
            mem = device_mem_capacity / 2 if mixed_precision else device_mem_capacity
            if mem >= 24000:
                big = output_height is not None and output_width is not None and int(output_height) * int(output_width) > 1920 * 1088
                tier = 2 if big else 1
            else:
                tier = 3 if mem >= 16000 else (4 if mem >= 8000 else 5)
        else:
            tier = vae_config + 2
        return {1: 0, 2: 1024, 3: 512, 4: 256}.get(tier, 128)

    # ---- layer graph (vae.py:338-369, 449-484) -------------------------------------------------------
    def _res(self, x, p, cache, idx):
        n = self.net
        h = n.conv(x, p + "shortcut") if (p + "shortcut") in n.convs else x
        y = x
        for gi, ci in (("0", "2"), ("3", "6")):
            y = n.norm(y, p + f"residual.{gi}.gamma")
            cx = _cache_update(y, cache[idx[0]])
            last = ci == "6"
            y = n.conv(y, p + f"residual.{ci}", cache=cache[idx[0]], res=h if last else None)
            cache[idx[0]] = cx
            idx[0] += 1
        return y

    def _cached_conv(self, x, name, cache, idx, **kw):
        cx = _cache_update(x, cache[idx[0]])
        y = self.net.conv(x, name, cache=cache[idx[0]], **kw)
        cache[idx[0]] = cx
        idx[0] += 1
        return y

    def _decoder(self, x, cache, idx):
        """Decoder3d.forward (vae.py:486-538) on one latent frame [1,h,w,32]."""
        n = self.net
        x = self._cached_conv(x, "decoder.conv1", cache, idx)
        x = self._res(x, "decoder.middle.0.", cache, idx)
        x = n.attention_block(x, "decoder.middle.1.")
        x = self._res(x, "decoder.middle.2.", cache, idx)
        li = 0
        tus = self.CFG["temperal_downsample"][::-1]
        for i in range(4):
            for _ in range(3):
                x = self._res(x, f"decoder.upsamples.{li}.", cache, idx); li += 1
            if i != 3:
                p = f"decoder.upsamples.{li}."
                if tus[i]:                                   # upsample3d (vae.py:151-189)
                    j = idx[0]
                    if cache[j] is None:
                        cache[j] = "Rep"
                    else:
                        prev = None if isinstance(cache[j], str) else cache[j]
                        cx = _cache_update(x, prev)
                        x = n.conv(x, p + "time_conv", cache=prev, interleave=True, pad_s=0)
                        cache[j] = cx
                    idx[0] += 1
                x = n.conv(x, p + "resample.1", ups=True)    # nearest-exact 2x + Conv2d 3x3 (vae.py:126-128)
                li += 1
        x = n.norm(x, "decoder.head.0.gamma")
        return self._cached_conv(x, "decoder.head.2", cache, idx, out_f32=True)

    def _encoder(self, x, cache, idx):
        """Encoder3d.forward (vae.py:371-428) on a chunk [t,H,W,32]."""
        n = self.net
        x = self._cached_conv(x, "encoder.conv1", cache, idx)
        li = 0
        tds = self.CFG["temperal_downsample"]
        for i in range(4):
            for _ in range(2):
                x = self._res(x, f"encoder.downsamples.{li}.", cache, idx); li += 1
            if i != 3:
                p = f"encoder.downsamples.{li}."
                x = n.conv(x, p + "resample.1", st_s=2, pad_s=0)          # ZeroPad2d((0,1,0,1)) + stride 2
                if tds[i]:                                   # downsample3d (vae.py:195-211)
                    j = idx[0]
                    if cache[j] is None:
                        cache[j] = x[-1:].clone()
                    else:
                        cx = x[-1:].clone()
                        prev2 = torch.cat([torch.zeros_like(cache[j]), cache[j]], 0)   # kernel reads cache[1] = last frame
                        x = n.conv(x, p + "time_conv", cache=prev2, st_t=2, front=1, pad_s=0)
                        cache[j] = cx
                    idx[0] += 1
                li += 1
        x = self._res(x, "encoder.middle.0.", cache, idx)
        x = n.attention_block(x, "encoder.middle.1.")
        x = self._res(x, "encoder.middle.2.", cache, idx)
        x = n.norm(x, "encoder.head.0.gamma")
        return self._cached_conv(x, "encoder.head.2", cache, idx)

    def _n_cached(self, side):
        return sum(1 for k, c in self.net.convs.items() if k.startswith(side) and (c.k[0] == 3))

    # ---- WanVAE_.decode (vae.py:628-662) ----------------------------------------------------------------
    def _decode_frames(self, z, want_u8, want_f32):
        lib = self.net.lib
        z = z.to(device=self.device, dtype=torch.float32).contiguous()          # [16, t, h, w]
        C, t, h, w = z.shape
        if getattr(self, "native", None) is not None:
            return self.native.decode(z, want_u8, want_f32)
        zp = torch.empty(t, h, w, 32, dtype=self._adt, device=self.device)
        inv_std = (1.0 / self.scale[1]).contiguous()                             # z / scale[1] + scale[0]
        self._pack(z, zp, inv_std, self.scale[0].contiguous(), C, 32, t * h * w)
        x = self.net.conv(zp, "conv2")                                           # 1x1x1, 16 -> 16 (padded to 32)
        T_out = (t - 1) * 4 + 1
        H, W = h * 8, w * 8
        u8 = torch.empty(3, T_out, H, W, dtype=torch.uint8, device=self.device) if want_u8 else None
        f32 = torch.empty(3, T_out, H, W, dtype=torch.float32, device=self.device) if want_f32 else None
        cache = [None] * self._n_cached("decoder.")
        t0 = 0
        for i in range(t):
            y = self._decoder(x[i:i + 1], cache, [0])                            # fp32 [T_i, H, W, 3]
            Ti = y.shape[0]
            check(lib.wan_vae_to_video(ptr(y), ptr(u8), ptr(f32), Ti, H * W, T_out, t0, stream_ptr()), "wan_vae_to_video")
            t0 += Ti
        assert t0 == T_out, (t0, T_out)
        return u8, f32

    # ---- spatial tiling (host logic of vae.py:664-717, :769-839, :841-881; latent-sized torch arithmetic on the seams) ------
    @staticmethod
    def _blend(a, b, be, dim):
        """blend_v / blend_h (vae.py:664-674), vectorised: b[y] = a[-be + y] * (1 - y/be) + b[y] * (y/be), Python-float weights."""
        be = min(a.shape[dim], b.shape[dim], be)
        if be <= 0:
            return b
        shape = [1] * b.dim()
        shape[dim] = be
        w = torch.arange(be, dtype=torch.float64) / be
        w0, w1 = (1 - w).to(torch.float32).view(shape).to(b.device), w.to(torch.float32).view(shape).to(b.device)
        bb = b.narrow(dim, 0, be)
        bb.copy_(a.narrow(dim, a.shape[dim] - be, be) * w0 + bb * w1)
        return b

    def _blend_tiles(self, rows, be, row_limit):
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(rows[i - 1][j], tile, be, -2)
                if j > 0:
                    tile = self._blend(row[j - 1], tile, be, -1)
                out.append(tile[..., :row_limit, :row_limit])
            out_rows.append(torch.cat(out, dim=-1))
        return torch.cat(out_rows, dim=-2)

    def _sharded_tiles(self, boxes, work, shape_of):
        """The tiles of a tiled decode / encode are independent until they are blended (vae.py:676-717, :769-839, :841-881) -- the
        unit of the multi-GPU VAE split: with `self.sp` set, tile k is computed by rank k % world and broadcast from it, so every
        rank holds every tile and blends them in the reference's order (replicated result, like the latents).  `boxes`: one
        (y, x) origin per tile; work(y, x) -> tile; shape_of(y, x) -> its shape (what a non-owner allocates).  Returns a function
        k -> tile; without `self.sp` it computes on demand (the streaming form keeps one tile alive at a time)."""
        sp = getattr(self, "sp", None)
        if sp is None or sp.world == 1:
            return lambda k: work(*boxes[k])
        import torch.distributed as dist
        grp = getattr(sp, "group", None)
        staged = dist.get_backend(grp) == "gloo" and self.device.type != "cpu"       # single-GPU functional tests: host staging
        tiles = []
        for k, (y, x) in enumerate(boxes):
            own = k % sp.world == sp.rank
            t = work(y, x).contiguous() if own else torch.empty(shape_of(y, x), dtype=torch.float32, device=self.device)
            tiles.append(t)
        for k, t in enumerate(tiles):                                                 # compute first, exchange after: the
            src = dist.get_global_rank(grp, k % sp.world) if grp is not None else k % sp.world   # owners run concurrently
            if staged:
                h = t.cpu()
                dist.broadcast(h, src=src, group=grp)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=grp)
        return lambda k: tiles[k]

    def _tiled_decode_f32(self, z, tile_size, end=False):
        """spatial_tiled_decode (vae.py:676-717) on one latent [16,t,h,w] -> fp32 [3,T,H,W] (not clamped)."""
        tl = int(tile_size / 8)
        ov, be = int(tl * 0.75), int(tile_size * self.upsampler_factor * 0.25)
        if tl < 1 or ov < 1:
            raise ValueError(f"tile_size {tile_size} is too small to tile")
        ys, xs = list(range(0, z.shape[-2], ov)), list(range(0, z.shape[-1], ov))
        get = self._sharded_tiles([(i, j) for i in ys for j in xs],
                                  lambda i, j: self._decode_clip(z[:, :, i:i + tl, j:j + tl], False, True, end)[1],
                                  lambda i, j: self._decoded_tile_shape(z, i, j, tl, end))
        rows = [[get(a * len(xs) + b) for b in range(len(xs))] for a in range(len(ys))]
        return self._blend_tiles(rows, be, tile_size * self.upsampler_factor - be)

    def _decoded_tile_shape(self, z, i, j, tl, end=False):
        f = 8 * self.upsampler_factor
        return (3, self._decoded_frames(z.shape[1], end), min(tl, z.shape[-2] - i) * f, min(tl, z.shape[-1] - j) * f)

    @staticmethod
    def _decoded_frames(t, end=False):
        return (t - 2) * 4 + 2 if (end and t > 1) else (t - 1) * 4 + 1

    def _decode_clip(self, z, want_u8, want_f32, any_end_frame=False):
        """_decode_frames, or with any_end_frame (vae.py:646-650) the clip without its last latent frame followed by that frame on
        its own: the reference runs it through the decoder with feat_cache=None -- no cache in, none out, no temporal upsampling
        -- which is exactly a one-frame clip (pinned on the oracle: tests/test_vae_oracle_vs_golden.py)."""
        if not any_end_frame or z.shape[1] < 2:
            return self._decode_frames(z, want_u8, want_f32)
        a, b = self._decode_frames(z[:, :-1], want_u8, want_f32), self._decode_frames(z[:, -1:], want_u8, want_f32)
        return tuple(None if x is None else torch.cat([x, y], dim=1) for x, y in zip(a, b))

    @staticmethod
    def _blend_edge(edge, tile, be, dim):
        """_blend_v_edge_ / _blend_h_edge_ (vae.py:23-40): the streaming form used by the tiled uint8 decode (fp32 tensor weights)."""
        be = min(int(edge.shape[dim]), int(tile.shape[dim]), int(be))
        if be <= 0:
            return
        shape = [1] * tile.dim()
        shape[dim] = be
        w = torch.arange(be, device=tile.device, dtype=tile.dtype).div_(be).view(shape)
        e = edge.narrow(dim, edge.shape[dim] - be, be).clone()
        e.mul_(1.0 - w)
        tile.narrow(dim, 0, be).mul_(w).add_(e)

    def _tiled_decode_u8(self, z, tile_size, end=False):
        """The tiled branch of decode_to_cpu_uint8 (vae.py:769-839) for the whole clip: uint8 [3,T,H,W] on the device."""
        tl = max(1, int(tile_size / 8))
        ov = max(1, int(tl * 0.75))
        be = int(tile_size * self.upsampler_factor * 0.25)
        row_limit = max(1, tile_size * self.upsampler_factor - be)
        T, H, W = self._decoded_frames(z.shape[1], end), z.shape[-2] * 8, z.shape[-1] * 8
        out = torch.empty(3, T, H, W, dtype=torch.uint8, device=self.device)
        boxes = []                                          # the tiles the loop below visits, in its order
        for r, ly in enumerate(range(0, z.shape[-2], ov)):
            if min(r * row_limit + row_limit, H) <= r * row_limit:
                break
            for c, lx in enumerate(range(0, z.shape[-1], ov)):
                if min(c * row_limit + row_limit, W) <= c * row_limit:
                    break
                boxes.append((ly, lx))
        get = self._sharded_tiles(boxes, lambda i, j: self._decode_clip(z[:, :, i:i + tl, j:j + tl], False, True, end)[1],
                                  lambda i, j: self._decoded_tile_shape(z, i, j, tl, end))
        prev_edges, r, k = [], 0, 0
        for ly in range(0, z.shape[-2], ov):
            y0, y1 = r * row_limit, min(r * row_limit + row_limit, H)
            if y1 <= y0:
                break
            cur_edges, left, c = [], None, 0
            for lx in range(0, z.shape[-1], ov):
                x0, x1 = c * row_limit, min(c * row_limit + row_limit, W)
                if x1 <= x0:
                    break
                tile = get(k)
                k += 1
                if r > 0 and c < len(prev_edges) and prev_edges[c] is not None:
                    self._blend_edge(prev_edges[c], tile, be, -2)
                if left is not None:
                    self._blend_edge(left, tile, be, -1)
                cur_edges.append(tile[..., -min(be, tile.shape[-2]):, :].clone() if y1 < H else None)
                left = tile[..., -min(be, tile.shape[-1]):].clone() if x1 < W else None
                tile = tile[..., :y1 - y0, :x1 - x0]
                out[:, :, y0:y0 + tile.shape[-2], x0:x0 + tile.shape[-1]] = \
                    tile.clamp(-1.0, 1.0).add_(1.0).mul_(127.5).round_().clamp_(0.0, 255.0).to(torch.uint8)    # vae.py:18-20
                c += 1
            prev_edges = cur_edges
            r += 1
        return out

    def decode(self, zs, tile_size=0, any_end_frame=False):
        end = bool(any_end_frame)
        if int(tile_size or 0) > 0:
            return [self._tiled_decode_f32(u.to(self.device), int(tile_size), end).clamp_(-1, 1) for u in zs]
        return [self._decode_clip(u, False, True, end)[1].clamp_(-1, 1) for u in zs]

    def decode_to_cpu_uint8(self, zs, tile_size=0, target_frames=None, target_height=None, target_width=None,
                            any_end_frame=False, frame_start=0):
        end = bool(any_end_frame)
        outs = []
        for u in zs:
            if int(tile_size or 0) > 0:
                u8 = self._tiled_decode_u8(u.to(self.device), int(tile_size), end)
            else:
                u8 = self._decode_clip(u, True, False, end)[0]
            T = u8.shape[1]
            fs = min(max(0, int(frame_start or 0)), T)
            te = T if target_frames is None else min(T, fs + int(target_frames))
            hh = u8.shape[2] if target_height is None else min(int(target_height), u8.shape[2])
            ww = u8.shape[3] if target_width is None else min(int(target_width), u8.shape[3])
            outs.append(u8[:, fs:te, :hh, :ww].to("cpu"))
        return outs

    # ---- WanVAE_.encode (vae.py:586-625) ------------------------------------------------------------------
    def encode(self, videos, tile_size=0, any_end_frame=False):
        if any_end_frame:
            # vae.py:590-606: 2 + (T - 2) // 4 chunks -- first frame, groups of four, and the clip's LAST frame through the encoder
            # with feat_cache=None, i.e. as a one-frame clip of its own (pinned on the oracle); per tile when tiling (:857)
            outs = []
            for v in videos:
                T = v.shape[1]
                if T < 2:
                    outs.append(self.encode([v], tile_size)[0])
                    continue
                body, last = self.encode([v[:, :1 + 4 * ((T - 2) // 4)], v[:, -1:]], tile_size)
                outs.append(torch.cat([body, last], dim=1))
            return outs
        if int(tile_size or 0) > 0:                          # spatial_tiled_encode (vae.py:841-881)
            ts = int(tile_size)
            tl = int(ts / 8)
            ov, be = int(ts * 0.75), int(tl * 0.25)
            outs = []
            for v in videos:
                v = v.to(self.device)
                ys, xs = list(range(0, v.shape[-2], ov)), list(range(0, v.shape[-1], ov))
                get = self._sharded_tiles([(i, j) for i in ys for j in xs],
                                          lambda i, j: self.encode([v[:, :, i:i + ts, j:j + ts]])[0],
                                          lambda i, j: (self.z_dim, (v.shape[1] - 1) // 4 + 1, min(ts, v.shape[-2] - i) // 8,
                                                        min(ts, v.shape[-1] - j) // 8))
                rows = [[get(a * len(xs) + b) for b in range(len(xs))] for a in range(len(ys))]
                outs.append(self._blend_tiles(rows, be, tl - be))   # blending commutes with the (affine) latent normalisation
            return outs
        lib = self.net.lib
        outs = []
        for v in videos:
            v = v.to(device=self.device, dtype=torch.float32).contiguous()       # [3, T, H, W]
            C, T, H, W = v.shape
            if getattr(self, "native", None) is not None:
                outs.append(self.native.encode(v))
                continue
            vp = torch.empty(T, H, W, 32, dtype=self._adt, device=self.device)
            self._pack(v, vp, None, None, C, 32, T * H * W)
            cache = [None] * self._n_cached("encoder.")
            chunks = []
            for i in range(1 + (T - 1) // 4):
                xc = vp[:1] if i == 0 else vp[1 + 4 * (i - 1):1 + 4 * i]
                chunks.append(self._encoder(xc, cache, [0]))
            enc = torch.cat(chunks, 0)                                           # [t, h, w, 32]
            mu = self.net.conv(enc, "conv1")                                     # 1x1x1 32 -> 32; mu = first 16
            t, h, w, _ = mu.shape
            out = torch.empty(16, t, h, w, dtype=torch.float32, device=self.device)
            self._unpack(mu, out, self.scale[0].contiguous(), self.scale[1].contiguous(), 16, 32, t * h * w)
            outs.append(out)
        return outs
