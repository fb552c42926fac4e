"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Wan DiT hot path.

This file is the *oracle*: a plain-torch, CPU, functional restatement of the arithmetic
the reference performs on the t2v / i2v2.2 path of ``WanModel.forward`` and of the
sampler loop around it.  It is never imported by the product package; it exists so that
``tests/`` (and ``bench.py``'s ``cpu_baseline`` leg, ``__graft_entry__.smoke()``) can
check the HIP path on the GPU box, where ``/root/reference`` does not exist.

Parity pin: the reference ships no tests/golden vectors for this path,
so this restatement is pinned against the reference's *own code* executed in the build
container (``oracle/ref_shim.py`` + ``oracle/make_golden.py``): in ``bf16`` mode every
function here reproduces the reference's eager CPU result bit-for-bit on the committed
fixtures (tests/test_oracle_vs_golden.py), and those fixtures travel to the GPU box.

Two precision plans:
  * ``dtype=torch.bfloat16`` -- the reference's real plan (model.py:1330-1371): bf16
    weights/activations, fp32 patch_embedding + head, fp32 RoPE, with every intermediate
    bf16 rounding the reference's in-place ops perform.
  * ``dtype=torch.float32``  -- same graph, fp32 everywhere: the accuracy anchor the
    HIP kernels (which keep fp32 longer than the reference) are measured against.

All file:line citations are into /root/reference.
Weights are a flat dict keyed by the checkpoint names (models/wan/convert_wan.py:19-76).
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration (models/wan/configs/*.json; SURVEY.md §8 shape table)
# --------------------------------------------------------------------------------------
@dataclass
class WanConfig:
    dim: int = 1536
    ffn_dim: int = 8960
    num_heads: int = 12
    num_layers: int = 30
    in_dim: int = 16          # 36 for i2v2_2 (model.py:1597-1600 concatenates y)
    out_dim: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    text_len: int = 512
    eps: float = 1e-6
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    model_type: str = "t2v"
    vace_layers: Optional[Tuple[int, ...]] = None   # VACE: main-block indices that carry a context block (model.py:1178-1206)
    vace_in_dim: int = 96
    flf: bool = False         # flf2v_720p: MLPProj(flf_pos_emb=True), clip_fea of TWO images (model.py:878-887, :1162)

    @property
    def head_dim(self):
        return self.dim // self.num_heads


CONFIGS = {
    "t2v_1.3B": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30),
    "t2v_14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
    "i2v2_2_14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=36, model_type="i2v2_2"),
    # test-sized configs (head_dim must stay 128: rope split 44/42/42, posemb_layers.py:356)
    "tiny": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2),
    "tiny_i2v": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, in_dim=36, model_type="i2v2_2"),
    "small": dict(dim=512, ffn_dim=1536, num_heads=4, num_layers=3),
    # Wan2.2 ti2v 5B (models/wan/configs/ti2v_2_2.json): 48-channel latents of the Wan2.2 VAE in and out, no y
    "ti2v_5B": dict(dim=3072, ffn_dim=14336, num_heads=24, num_layers=30, in_dim=48, out_dim=48, model_type="ti2v2_2"),
    "tiny_ti2v": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, in_dim=48, out_dim=48, model_type="ti2v2_2"),
    # Wan2.1 i2v: CLIP image tokens through img_emb + the k_img / v_img cross-attention branch (model.py:448-499, :868-889)
    "i2v_14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=36, model_type="i2v"),
    "tiny_i2v21": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, in_dim=36, model_type="i2v"),
    # Wan2.1 flf2v 720p (first + last frame): the i2v model with a position embedding on the CLIP tokens of two images; the second
    # image's 257 tokens land in front of the TEXT tokens of the cross-attention (WanI2VCrossAttention splits at 257, model.py:472-473)
    "flf2v_14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=36, model_type="i2v", flf=True),
    "tiny_flf2v": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, in_dim=36, model_type="i2v", flf=True),
    # VACE (Wan2.1 VACE 14B: vace_layers 0,5,...,35; 1.3B: every second block): context blocks feeding hints into the main blocks
    "vace_14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, vace_layers=tuple(range(0, 40, 5))),
    "tiny_vace": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=3, vace_layers=(0, 2)),
}
CLIP_TOKENS, CLIP_DIM = 257, 1280


def make_config(name: str) -> WanConfig:
    return WanConfig(**CONFIGS[name])


# --------------------------------------------------------------------------------------
# synthetic weights (N(0,.02) linears, N(0,.01) biases, modulation N(0,1)/sqrt(d))
# --------------------------------------------------------------------------------------
def param_shapes(cfg: WanConfig) -> Dict[str, Tuple[int, ...]]:
    """Checkpoint key -> shape (model.py:1131-1160, 509-556, 831-845)."""
    d, f = cfg.dim, cfg.ffn_dim
    p = {}
    p["patch_embedding.weight"] = (d, cfg.in_dim, *cfg.patch_size)
    p["patch_embedding.bias"] = (d,)
    p["text_embedding.0.weight"] = (d, cfg.text_dim); p["text_embedding.0.bias"] = (d,)
    p["text_embedding.2.weight"] = (d, d); p["text_embedding.2.bias"] = (d,)
    p["time_embedding.0.weight"] = (d, cfg.freq_dim); p["time_embedding.0.bias"] = (d,)
    p["time_embedding.2.weight"] = (d, d); p["time_embedding.2.bias"] = (d,)
    p["time_projection.1.weight"] = (6 * d, d); p["time_projection.1.bias"] = (6 * d,)
    for i in range(cfg.num_layers):
        b = f"blocks.{i}."
        p[b + "modulation"] = (1, 6, d)
        for a in ("self_attn", "cross_attn"):
            for l in ("q", "k", "v", "o"):
                p[b + f"{a}.{l}.weight"] = (d, d); p[b + f"{a}.{l}.bias"] = (d,)
            p[b + f"{a}.norm_q.weight"] = (d,); p[b + f"{a}.norm_k.weight"] = (d,)
        p[b + "norm3.weight"] = (d,); p[b + "norm3.bias"] = (d,)
        p[b + "ffn.0.weight"] = (f, d); p[b + "ffn.0.bias"] = (f,)
        p[b + "ffn.2.weight"] = (d, f); p[b + "ffn.2.bias"] = (d,)
    if cfg.vace_layers is not None:                            # VaceWanAttentionBlock (:790-828), vace_patch_embedding (:1203-1206)
        for n in range(len(cfg.vace_layers)):
            b = f"vace_blocks.{n}."
            p[b + "modulation"] = (1, 6, d)
            for a in ("self_attn", "cross_attn"):
                for l in ("q", "k", "v", "o"):
                    p[b + f"{a}.{l}.weight"] = (d, d); p[b + f"{a}.{l}.bias"] = (d,)
                p[b + f"{a}.norm_q.weight"] = (d,); p[b + f"{a}.norm_k.weight"] = (d,)
            p[b + "norm3.weight"] = (d,); p[b + "norm3.bias"] = (d,)
            p[b + "ffn.0.weight"] = (f, d); p[b + "ffn.0.bias"] = (f,)
            p[b + "ffn.2.weight"] = (d, f); p[b + "ffn.2.bias"] = (d,)
            if n == 0:
                p[b + "before_proj.weight"] = (d, d); p[b + "before_proj.bias"] = (d,)
            p[b + "after_proj.weight"] = (d, d); p[b + "after_proj.bias"] = (d,)
        p["vace_patch_embedding.weight"] = (d, cfg.vace_in_dim, *cfg.patch_size)
        p["vace_patch_embedding.bias"] = (d,)
    if cfg.model_type == "i2v":                                # WanI2VCrossAttention (:448-464) + MLPProj (:868-876)
        for i in range(cfg.num_layers):
            b = f"blocks.{i}.cross_attn."
            p[b + "k_img.weight"] = (d, d); p[b + "k_img.bias"] = (d,)
            p[b + "v_img.weight"] = (d, d); p[b + "v_img.bias"] = (d,)
            p[b + "norm_k_img.weight"] = (d,)
        p["img_emb.proj.0.weight"] = (CLIP_DIM,); p["img_emb.proj.0.bias"] = (CLIP_DIM,)
        p["img_emb.proj.1.weight"] = (CLIP_DIM, CLIP_DIM); p["img_emb.proj.1.bias"] = (CLIP_DIM,)
        p["img_emb.proj.3.weight"] = (d, CLIP_DIM); p["img_emb.proj.3.bias"] = (d,)
        p["img_emb.proj.4.weight"] = (d,); p["img_emb.proj.4.bias"] = (d,)
        if cfg.flf:
            p["img_emb.emb_pos"] = (1, 2 * CLIP_TOKENS, CLIP_DIM)   # MLPProj(flf_pos_emb=True) (:878-881)
    p["head.modulation"] = (1, 2, d)
    p["head.head.weight"] = (cfg.out_dim * math.prod(cfg.patch_size), d)
    p["head.head.bias"] = (cfg.out_dim * math.prod(cfg.patch_size),)
    return p


FP32_LOCKED = ("patch_embedding.", "head.")  # model.py:1331 (lock_layers_dtypes layer_list)
# `mixed_precision_transformer` (wgp.py:4039 server setting "mixed_precision" -> any2video.py:190 lock_layers_dtypes(torch.float32)):
# layer_list2 of model.py:1338-1345 is kept in fp32 as well -- the time MLP, the time projection and every block's norm3.  The weights are
# the checkpoint's bf16 values (upcast, exact); what changes is the arithmetic that follows from them by type promotion: the modulation
# dtype (= time_projection[1].weight.dtype, model.py:1545) becomes fp32, so the residual stream x, e / e0 and every AdaLN modulate /
# gated residual run in fp32, with ONE rounding to bf16 in front of each Linear / attention (`.to(attention_dtype)`, model.py:650,665,692)
MIXED_LOCKED = ("time_embedding.", "time_projection.")


def is_fp32_locked(key: str, mixed: bool = False) -> bool:
    """Does this parameter stay fp32 under the reference's dtype locks (model.py:1330-1371)?"""
    return key.startswith(FP32_LOCKED) or (mixed and (key.startswith(MIXED_LOCKED) or ".norm3." in key))


def synth_weights(cfg: WanConfig, seed: int = 1234, dtype=torch.bfloat16, max_layers: Optional[int] = None, mixed: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic checkpoint. fp32 master values are drawn first, then cast per the
    reference's dtype lock (patch_embedding/head fp32, rest `dtype`), so the bf16 and fp32
    plans share the *same* (bf16-representable) weights: the master is rounded through
    bf16 for every non-locked tensor."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if max_layers is not None and k.startswith(f"blocks.{max_layers}."):
            break       # a PREFIX of the checkpoint (embeddings + the first blocks, same values; no head): cheap partial checks
        if k.endswith("modulation"):
            w = torch.randn(shp, generator=g) / cfg.dim ** 0.5
        elif ("norm" in k or k in ("img_emb.proj.0.weight", "img_emb.proj.4.weight")) and k.endswith("weight"):
            w = 1.0 + 0.02 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            w = 0.01 * torch.randn(shp, generator=g)
        else:
            w = 0.02 * torch.randn(shp, generator=g)
        if not k.startswith(FP32_LOCKED):
            w = w.to(torch.bfloat16).to(torch.float32)  # bf16-representable master
            if not is_fp32_locked(k, mixed):            # mixed: the same bf16 values, held in fp32 (the loader's upcast is exact)
                w = w.to(dtype)
        out[k] = w
    return out


# --------------------------------------------------------------------------------------
# RoPE tables -- posemb_layers.py:346-431 (get_nd_rotary_pos_embed), :434-476, :492-525
# --------------------------------------------------------------------------------------
ROPE_DIM_LIST = (44, 42, 42)  # posemb_layers.py:356 (t, h, w) for head_dim 128


def rope_tables(grid: Sequence[int], theta: float = 10000.0, riflex: bool = False, riflex_k: int = 6) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [L,128] fp32 for a token grid (f, h/2, w/2), f-major token order.
    Follows get_rotary_pos_embed -> get_nd_rotary_pos_embed -> get_1d_rotary_pos_embed:
    positions are integer grid indices in fp32 (USE_FP32_ROPE_FREQS=True, :6),
    freqs_i = 1/theta^(2i/dim_axis) in fp32, angles = outer(pos, freqs) fp32,
    cos/sin repeat_interleave(2)."""
    f, h, w = [int(v) for v in grid]
    axes = torch.meshgrid(torch.arange(f, dtype=torch.float32), torch.arange(h, dtype=torch.float32),
                          torch.arange(w, dtype=torch.float32), indexing="ij")
    cos_parts, sin_parts = [], []
    for ax, (dim_axis, pos) in enumerate(zip(ROPE_DIM_LIST, axes)):
        pos = pos.reshape(-1)
        freqs = 1.0 / (theta ** (torch.arange(0, dim_axis, 2, dtype=torch.float32)[: dim_axis // 2] / dim_axis))
        if ax == 0 and riflex:                              # get_1d_rotary_pos_embed_riflex (posemb_layers.py:70-76), L_test = f
            freqs[riflex_k - 1] = 0.9 * 2 * torch.pi / f
        ang = torch.outer(pos, freqs)
        cos_parts.append(ang.cos().repeat_interleave(2, dim=1))
        sin_parts.append(ang.sin().repeat_interleave(2, dim=1))
    return torch.cat(cos_parts, dim=1), torch.cat(sin_parts, dim=1)


def rope_apply(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B,L,H,128]; posemb_layers.py:251-269: up-cast to fp32, rotate adjacent pairs,
    one rounding back to x.dtype."""
    xw = x.to(torch.float32)
    c = cos.view(1, cos.shape[0], 1, -1, 2)
    s = sin.view(1, sin.shape[0], 1, -1, 2)
    xv = xw.view(*xw.shape[:-1], -1, 2)
    x0, x1 = xv[..., 0], xv[..., 1]
    # x0.mul_(cos0).addcmul_(x1, sin0, value=-1); x1.mul_(cos1).addcmul_(x0_orig, sin1)
    o0 = torch.addcmul(x0 * c[..., 0], x1, s[..., 0], value=-1)
    o1 = torch.addcmul(x1 * c[..., 1], x0, s[..., 1])
    return torch.stack([o0, o1], dim=-1).flatten(-2).to(x.dtype)


# --------------------------------------------------------------------------------------
# norms -- model.py:152-175 (WanRMSNorm), :194-213 (WanLayerNorm)
# --------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """Full-width RMSNorm with the reference's two roundings (model.py:165-174):
    y = rsqrt(mean(x.float()^2) + eps) fp32; x *= y (rounded to x.dtype); x *= weight."""
    y = x.float().pow(2).mean(dim=-1, keepdim=True)
    y = (y + eps).rsqrt()
    x = (x * y).to(x.dtype)  # in-place `x *= y`: bf16 tensor x fp32 scalar-per-row -> bf16
    return x * weight.to(x.dtype) if weight.dtype != x.dtype else x * weight


def layer_norm(x: torch.Tensor, eps: float, weight=None, bias=None) -> torch.Tensor:
    """WanLayerNorm.forward (model.py:199-212): F.layer_norm in x's dtype (fp32 stats
    inside), optional affine computed in the weight's dtype then cast back."""
    if weight is not None:
        y = F.layer_norm(x.to(weight.dtype), (x.shape[-1],), weight, bias, eps)
        return y.type_as(x)
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


# --------------------------------------------------------------------------------------
# attention -- shared/attention.py:208-225 (sdpa_wrapper), :402-416, :563
# --------------------------------------------------------------------------------------
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, exact: bool = False) -> torch.Tensor:
    """q [B,Lq,H,D], k/v [B or 1,Lk,H,D] -> [B,Lq,H,D]; softmax(QK^T/sqrt(D))V, no mask.
    exact=False: torch SDPA in the input dtype, exactly the reference's default "sdpa" mode.
    exact=True : fp32 softmax/accumulate on fp32 copies (backend-independent anchor)."""
    b = q.shape[0]
    if k.shape[0] != b:
        k = k.expand(b, -1, -1, -1)
    if v.shape[0] != b:
        v = v.expand(b, -1, -1, -1)
    if exact:
        qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
        s = torch.matmul(qf, kf.transpose(-1, -2)) / math.sqrt(q.shape[-1])
        o = torch.matmul(torch.softmax(s, dim=-1), vf)
        return o.transpose(1, 2).to(q.dtype)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return o.transpose(1, 2)


def _linear(x, W, prefix):
    w = W[prefix + ".weight"]
    if w.dtype == torch.float8_e4m3fn:                       # scaled-fp8 checkpoint: QLinearScaledFP8 (shared/qtypes/scaled_fp8.py)
        from . import fp8_oracle
        return fp8_oracle.linear_scaled(x, w, W[prefix + ".scale_weight"], W[prefix + ".bias"])
    if w.dtype == torch.float8_e5m2:                         # scaled_float8_e5m2: torch._scaled_mm takes no e5m2 x e5m2 product (the probe at
        from . import fp8_oracle                             # scaled_fp8.py:197-221 fails), every such Linear runs _linear_fallback (:306-322)
        return fp8_oracle.linear_fallback(x, w, W[prefix + ".scale_weight"], W[prefix + ".bias"], dtype=x.dtype)
    return F.linear(x, w, W[prefix + ".bias"])


FP8_BLOCK_LINEARS = tuple(f"{a}.{l}" for a in ("self_attn", "cross_attn") for l in "qkvo") + ("ffn.0", "ffn.2")


def quantize_checkpoint_fp8(W, per_row: bool = True, fp8_dtype=None):
    """A scaled-fp8 checkpoint from a bf16 one: the ten Linears of every block become float8_e4m3fn `.weight` + fp32
    `.scale_weight` (the layout QLinearScaledFP8._load_from_state_dict reads, scaled_fp8.py:563-637); everything else
    (embeddings, norms, modulation, head, biases) stays as it is."""
    from . import fp8_oracle
    out = dict(W)
    for k in list(W):
        if k.startswith(("blocks.", "vace_blocks.")) and k.endswith(".weight") and k[:-7].split(".", 2)[2] in FP8_BLOCK_LINEARS:
            q, s_ = fp8_oracle.quantize_weight(W[k].float(), per_row=per_row, fp8_dtype=fp8_dtype)
            out[k] = q
            out[k[:-7] + ".scale_weight"] = s_.float()
    return out


# --------------------------------------------------------------------------------------
# embeddings -- model.py:32-42, :1815-1818, :1856
# --------------------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    half = dim // 2
    position = position.type(torch.float32)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def time_embed(t: torch.Tensor, W, cfg: WanConfig, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """e [B,dim], e0 [B,6,dim] (model.py:1815-1818); modulation dtype = time_projection dtype."""
    x = sinusoidal_embedding_1d(cfg.freq_dim, t.flatten()).to(dtype)
    e = _linear(F.silu(_linear(x, W, "time_embedding.0")), W, "time_embedding.2")
    e0 = _linear(F.silu(e), W, "time_projection.1").unflatten(1, (6, cfg.dim)).to(e.dtype)
    return e, e0


def text_embed(ctx: torch.Tensor, W) -> torch.Tensor:
    """model.py:1133-1135,1856: Linear -> GELU(tanh) -> Linear on [B,512,4096]."""
    return _linear(F.gelu(_linear(ctx, W, "text_embedding.0"), approximate="tanh"), W, "text_embedding.2")


def patch_embed(x: torch.Tensor, W, cfg: WanConfig, dtype) -> Tuple[torch.Tensor, Tuple[int, int, int]]:
    """model.py:1631,1731: fp32 Conv3d k=s=(1,2,2) -> .to(modulation dtype) -> [B,L,dim]."""
    w = W["patch_embedding.weight"]
    if x.is_cuda:
        # the oracle executed on a GPU (tests/test_gpu_14B_depth.py): kernel == stride, so the convolution is one matmul over
        # the gathered patches -- no vendor convolution library; the CPU path below is the one pinned to the reference
        pt, ph, pw = cfg.patch_size
        B, C, Fr, H, Wd = x.shape
        g = (Fr // pt, H // ph, Wd // pw)
        xp = x.to(w.dtype).reshape(B, C, g[0], pt, g[1], ph, g[2], pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, g[0] * g[1] * g[2], -1)
        y = (xp @ w.reshape(w.shape[0], -1).t() + W["patch_embedding.bias"]).to(dtype)
        return y, g
    y = F.conv3d(x.to(w.dtype), w, W["patch_embedding.bias"], stride=cfg.patch_size).to(dtype)
    grid = tuple(y.shape[2:])
    return y.flatten(2).transpose(1, 2), grid


# --------------------------------------------------------------------------------------
# block -- model.py:575-724 (t2v path), self-attn :309-407, cross-attn :245-265,410-445
# --------------------------------------------------------------------------------------
def self_attention(x, W, p, cfg: WanConfig, cos, sin, exact):
    b, s, n, d = x.shape[0], x.shape[1], cfg.num_heads, cfg.head_dim
    q = _linear(x, W, p + "q"); k = _linear(x, W, p + "k"); v = _linear(x, W, p + "v")
    q = rms_norm(q, W[p + "norm_q.weight"], cfg.eps)
    k = rms_norm(k, W[p + "norm_k.weight"], cfg.eps)
    q, k, v = q.view(b, s, n, d), k.view(b, s, n, d), v.view(b, s, n, d)
    q, k = rope_apply(q, cos, sin), rope_apply(k, cos, sin)
    o = attention(q, k, v, exact)
    return _linear(o.flatten(2), W, p + "o")


def nag_combine(x_pos, x_neg, nag_scale, nag_tau, nag_alpha):
    """Normalized attention guidance on the two cross-attention results (text_cross_attention, model.py:276-293), one torch
    statement per reference statement on the tensors' own dtype (bf16 in the real plan: every in-place op rounds):
    guidance = x_neg (1 - s) + s x_pos; rows whose L1 norm exceeds tau times the positive row's are scaled back to tau;
    result = alpha guidance + (1 - alpha) x_pos."""
    x_pos, x_neg = x_pos.clone(), x_neg.clone()
    x_neg.mul_(1 - nag_scale)                                                  # :278
    x_neg.add_(x_pos, alpha=nag_scale)                                         # :279
    x_guidance = x_neg
    norm_positive = torch.norm(x_pos, p=1, dim=-1, keepdim=True)               # :282
    norm_guidance = torch.norm(x_guidance, p=1, dim=-1, keepdim=True)          # :283
    scale = torch.nan_to_num(norm_guidance / norm_positive, 10)                # :284-285
    factor = 1 / (norm_guidance + 1e-7) * norm_positive * nag_tau              # :286
    x_guidance = torch.where(scale > nag_tau, x_guidance * factor, x_guidance)  # :287
    x_pos.mul_(1 - nag_alpha)                                                  # :289
    x_guidance.mul_(nag_alpha)                                                 # :290
    x_guidance.add_(x_pos)                                                     # :291
    return x_guidance


def cross_attention(x, ctx, W, p, cfg: WanConfig, exact, nag=None):
    """nag = (nag_scale, nag_tau, nag_alpha) = offload.shared_state["_nag_*"] (any2video.py:607): with nag_scale > 1 a context
    of batch 2 is (positive ; negative) prompt and the two attention results go through nag_combine (model.py:260-292)."""
    b, n, d = x.shape[0], cfg.num_heads, cfg.head_dim
    q = rms_norm(_linear(x, W, p + "q"), W[p + "norm_q.weight"], cfg.eps).view(b, -1, n, d)
    ctx_img = None
    if cfg.model_type == "i2v":                                # WanI2VCrossAttention.forward (model.py:466-499)
        ctx_img, ctx = ctx[:, :CLIP_TOKENS], ctx[:, CLIP_TOKENS:]
        ctx_img = ctx_img[:b]                                  # :476-477
    k = rms_norm(_linear(ctx, W, p + "k"), W[p + "norm_k.weight"], cfg.eps).view(ctx.shape[0], -1, n, d)
    v = _linear(ctx, W, p + "v").view(ctx.shape[0], -1, n, d)
    if nag is not None and nag[0] > 1 and k.shape[0] != 1:     # :260 `nag_scale <= 1 or len(k)==1` is the plain path
        o = nag_combine(attention(q, k[:1], v[:1], exact).flatten(2, 3), attention(q, k[1:], v[1:], exact).flatten(2, 3), *nag)
    else:
        o = attention(q, k, v, exact).flatten(2, 3)
    if ctx_img is not None:
        k_img = rms_norm(_linear(ctx_img, W, p + "k_img"), W[p + "norm_k_img.weight"], cfg.eps).view(ctx_img.shape[0], -1, n, d)
        v_img = _linear(ctx_img, W, p + "v_img").view(ctx_img.shape[0], -1, n, d)
        o = o + attention(q, k_img, v_img, exact).flatten(2, 3)        # x += img_x (:493)
    return _linear(o, W, p + "o")


def img_emb(clip_fea, W):
    """MLPProj.forward (model.py:868-889): [flf2v: the two images' tokens as one sequence + the position embedding (:884-887)] ->
    LayerNorm(1280) -> Linear -> GELU(erf) -> Linear -> LayerNorm(dim); torch.nn.LayerNorm's default eps 1e-5."""
    if "img_emb.emb_pos" in W:
        bs, n, d = clip_fea.shape
        clip_fea = clip_fea.reshape(-1, 2 * n, d).to(W["img_emb.emb_pos"].dtype) + W["img_emb.emb_pos"]
    x = F.layer_norm(clip_fea, (clip_fea.shape[-1],), W["img_emb.proj.0.weight"], W["img_emb.proj.0.bias"], 1e-5)
    x = F.gelu(_linear(x, W, "img_emb.proj.1"))
    x = _linear(x, W, "img_emb.proj.3")
    return F.layer_norm(x, (x.shape[-1],), W["img_emb.proj.4.weight"], W["img_emb.proj.4.bias"], 1e-5)


def block_forward(x, e0, ctx, cos, sin, W, i, cfg: WanConfig, exact: bool = False, nag=None, adt=None):
    """WanAttentionBlock.forward, t2v path (model.py:631-711).  x [B,L,dim] in the
    residual dtype, e0 [1,6,dim] (latent_frames = e.shape[0] = 1, so the reshape at
    :635/:658/:688 is a no-op broadcast).  i: block index, or a key prefix (VACE context blocks)."""
    p = i if isinstance(i, str) else f"blocks.{i}."
    nf = e0.shape[0]                                           # latent_frames (:631): > 1 with per-frame timesteps (ti2v injection,
    rs = (lambda v: v.reshape(v.shape[0], nf, -1, v.shape[-1])) if nf > 1 else (lambda v: v)      # diffusion forcing): reshape_latent
    un = (lambda v: v.reshape(v.shape[0], -1, v.shape[-1])) if nf > 1 else (lambda v: v)          # / restore_latent_shape (:45-49)
    # adt: the attention dtype (model.py:614) where it differs from the residual stream's -- the mixed-precision plan (x, e0 fp32; the
    # Linears bf16): one rounding in front of every Linear / attention (:650, :665, :692), results widened back (:654, :668, :708).
    # None: the two are the same dtype and every cast below is the identity (the bf16 plan, the fp32 anchor).
    dt = x.dtype
    to_a = (lambda v: v.to(adt)) if adt is not None else (lambda v: v)
    e = (W[p + "modulation"] + e0).chunk(6, dim=1)            # :632   6 x [nf,1,dim]
    x_mod = rs(layer_norm(x, cfg.eps))                         # :634-635
    x_mod = x_mod * (1 + e[1]); x_mod = un(x_mod + e[0])       # :636-638 (two roundings)
    y = self_attention(to_a(x_mod), W, p + "self_attn.", cfg, cos, sin, exact).to(dt)   # :650-654
    x = un(torch.addcmul(rs(x), rs(y), e[2]))                  # :658-660
    y = layer_norm(x, cfg.eps, W[p + "norm3.weight"], W[p + "norm3.bias"])  # :664
    x = x + cross_attention(to_a(y), ctx, W, p + "cross_attn.", cfg, exact, nag).to(dt)  # :665-668
    y = rs(layer_norm(x, cfg.eps))                             # :686-688
    y = y * (1 + e[4]); y = to_a(un(y + e[3]))                 # :689-692
    shp = y.shape                                              # :698-707 three row chunks
    y2 = y.reshape(-1, shp[-1])
    outs = [_linear(F.gelu(_linear(c, W, p + "ffn.0"), approximate="tanh"), W, p + "ffn.2")
            for c in torch.split(y2, int(y2.shape[0] / 2.7))]
    y = torch.cat(outs, 0).view(shp).to(dt)                    # :708
    x = un(torch.addcmul(rs(x), rs(y), e[5]))                  # :708-711
    return x


def vace_block_forward(c, x, e0, ctx, cos, sin, W, n: int, cfg: WanConfig, exact: bool = False, nag=None):
    """VaceWanAttentionBlock.forward (model.py:816-828) as called from the main block (:617-629): returns (c, c_skip)."""
    p = f"vace_blocks.{n}."
    if n == 0:
        c = _linear(c, W, p + "before_proj")
        c = c + x                                            # c += x
    c = block_forward(c, e0, ctx, cos, sin, W, p, cfg, exact, nag)
    return c, _linear(c, W, p + "after_proj")


def vace_hints(vace_context, vace_scale, W, cfg: WanConfig, n_streams: int):
    """model.py:1905-1912: every VACE context (one tensor, or a list of them with one scale each) is embedded once; every x stream
    gets its own copy of every context's hint tokens.  Returns (hints[stream][context], scales) or (None, None)."""
    if vace_context is None:
        return None, None
    ctxs = list(vace_context) if isinstance(vace_context, (list, tuple)) else [vace_context]
    scales = list(vace_scale) if isinstance(vace_scale, (list, tuple)) else [vace_scale] * len(ctxs)
    w = W["vace_patch_embedding.weight"]
    emb = [F.conv3d(u.to(w.dtype).unsqueeze(0), w, W["vace_patch_embedding.bias"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
           for u in ctxs]
    return [[c.clone() for c in emb] for _ in range(n_streams)], scales


def block_with_hints(x, hints, scales, e0, ctx, cos, sin, W, i: int, cfg: WanConfig, exact: bool = False, nag=None, adt=None):
    """One main block with its VACE context block(s) (model.py:617-629 in front of the block, :713-719 behind it): every context
    with a non-zero scale runs the context block on its own hint stream (in place in `hints`); the projected hints are added to
    x in context order, each add rounding to the stream's dtype."""
    skips = []
    if hints is not None and cfg.vace_layers is not None and i in cfg.vace_layers:
        n = cfg.vace_layers.index(i)
        for k, sc in enumerate(scales):
            if sc == 0:
                skips.append(None)
                continue
            hints[k], sk = vace_block_forward(hints[k], x, e0, ctx, cos, sin, W, n, cfg, exact, nag)
            skips.append(sk)
    x = block_forward(x, e0, ctx, cos, sin, W, i, cfg, exact, nag, adt)
    for sk, sc in zip(skips, scales or []):
        if sk is not None:
            x = x + sk if sc == 1 else torch.add(x, sk, alpha=sc)
    return x


# --------------------------------------------------------------------------------------
# head + unpatchify -- model.py:847-865, :2100-2126
# --------------------------------------------------------------------------------------
def head_forward(x, e, W, cfg: WanConfig):
    dtype = x.dtype
    nf = e.shape[0]                                                 # latent_frames (model.py:856)
    em = (W["head.modulation"] + e.unsqueeze(1)).chunk(2, dim=1)   # fp32 (+ bf16 e -> fp32)
    x = layer_norm(x, cfg.eps).to(dtype)
    if nf > 1:
        x = x.reshape(x.shape[0], nf, -1, x.shape[-1])
    x = x * (1 + em[1])     # `x *= (1+e[1])` in place on the bf16 tensor -> rounds to bf16 ...
    x = x.to(dtype)
    x = x + em[0]
    x = x.to(dtype)         # ... and `x += e[0]` likewise (model.py:860-861)
    if nf > 1:
        x = x.reshape(x.shape[0], -1, x.shape[-1])
    x = x.to(W["head.head.weight"].dtype)
    return F.linear(x, W["head.head.weight"], W["head.head.bias"])


def unpatchify(x, grid, cfg: WanConfig):
    """[B,L,out*prod(patch)] -> [B,out,F,H,W]  ('fhwpqrc->cfphqwr', model.py:2119-2121)."""
    c = cfg.out_dim
    outs = []
    for u in x:
        u = u[: math.prod(grid)].view(*grid, *cfg.patch_size, c)
        u = torch.einsum("fhwpqrc->cfphqwr", u)
        outs.append(u.reshape(c, *[i * j for i, j in zip(grid, cfg.patch_size)]))
    return torch.stack(outs, 0)


# --------------------------------------------------------------------------------------
# WanModel.forward -- model.py:1485-2098 (t2v / i2v2_2 path, cache None)
# --------------------------------------------------------------------------------------
def dit_forward(x_list: List[torch.Tensor], t: torch.Tensor, context_list: List[torch.Tensor],
                W, cfg: WanConfig, y: Optional[torch.Tensor] = None, freqs=None,
                dtype=torch.bfloat16, exact: bool = False, return_hidden: bool = False, clip_fea: Optional[torch.Tensor] = None,
                vace_context=None, vace_scale=1.0, probe=None, nag=None, perturbation_layers=None):
    """x_list: S tensors [B,16,F,H,W] fp32; t [1]; context_list: S tensors [B,512,4096] ([2,512,4096] = positive ; negative
    prompt for a stream under normalized attention guidance, nag = (scale, tau, alpha), any2video.py:607-608).
    perturbation_layers: block indices every stream but the first passes through unchanged (skip-layer guidance,
    any2video.py:1502, model.py:2025-2028; joint pass, x_id 0).
    Returns S fp32 tensors [B,16,F,H,W] (model.py:2093-2097).
    probe(i, s, hidden): called with stream s's token stream after block i (error-growth tables)."""
    hs = []
    grid = None
    # modulation dtype = time_projection[1].weight.dtype (model.py:1545): `dtype` in the bf16 plan and in the fp32 anchor, fp32 under the
    # mixed-precision locks (synth_weights(mixed=True)) -- then the residual stream runs in fp32 between bf16 Linears (block_forward `adt`)
    mdt = W["time_projection.1.weight"].dtype
    adt = dtype if mdt != dtype else None
    if adt is not None and vace_context is not None:
        raise NotImplementedError("wan_oracle: the mixed-precision plan is restated for the t2v / i2v2_2 / ti2v / i2v (CLIP) block chain, not for VACE")
    for x in x_list:
        if y is not None:                                   # model.py:1597-1600
            yy = y.unsqueeze(0)
            if x.shape[0] > 1:
                yy = yy.expand(x.shape[0], -1, -1, -1, -1)
            x = torch.cat([x, yy.to(x.dtype)], dim=1)
        h, grid = patch_embed(x, W, cfg, mdt)
        hs.append(h)
    cos, sin = freqs if freqs is not None else rope_tables(grid)
    e, e0 = time_embed(t, W, cfg, mdt)
    ctxs = [text_embed(c.to(dtype), W) for c in context_list]
    if clip_fea is not None:                                # model.py:1858-1869: [clip tokens ; text tokens]
        cc = img_emb(clip_fea.to(dtype), W)
        ctxs = [torch.cat([cc.repeat(len(c), 1, 1) if len(c) != len(cc) else cc, c], dim=1) for c in ctxs]   # :1864-1868
    hints, scales = vace_hints(vace_context, vace_scale, W, cfg, len(hs))
    for i in range(cfg.num_layers):                         # model.py:1993-2036
        for s in range(len(hs)):
            if perturbation_layers is not None and i in perturbation_layers and s != 0:
                continue                                    # skip-layer guidance (:2025-2028): only the first stream runs such a block
            hs[s] = block_with_hints(hs[s], None if hints is None else hints[s], scales, e0, ctxs[s], cos, sin, W, i, cfg, exact, nag, adt)
            if probe is not None:
                probe(i, s, hs[s])
    if return_hidden:
        return hs
    outs = []
    for h in hs:
        o = head_forward(h, e, W, cfg)
        outs.append(unpatchify(o, grid, cfg).float())
    return outs


# --------------------------------------------------------------------------------------
# schedulers -- shared/utils/fm_solvers_unipc.py, euler_scheduler.py
# --------------------------------------------------------------------------------------
class UniPCOracle:
    """FlowUniPCMultistepScheduler restated (bh2, order 2, predict_x0, flow_prediction,
    lower_order_final) -- fm_solvers_unipc.py:77-132 (init), :160-228 (set_timesteps),
    :279-348 (convert_model_output), :350-480 (UniP), :482-626 (UniC), :655-739 (step)."""

    def __init__(self, num_train_timesteps=1000, solver_order=2):
        self.N = num_train_timesteps
        self.order = solver_order

    def set_timesteps(self, num_inference_steps: int, shift: float):
        alphas = np.linspace(1, 1 / self.N, self.N)[::-1].copy()      # :113-121 (constructed with shift=1)
        s0 = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigma_min, sigma_max = s0[-1].item(), s0[0].item()            # :130-131  (0.0, fp32(0.999))
        sigmas = np.linspace(sigma_max, sigma_min, num_inference_steps + 1).copy()[:-1]   # :186-188
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                                 # :196-197
        timesteps = sigmas * self.N
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)                            # :210-211
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)                  # :214-215
        self.model_outputs = [None] * self.order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = 1
        return self.timesteps

    def _lam(self, sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _coefs(self, sig_t, sig_s0, order, rk_sigma):
        lam_t, lam_s0 = self._lam(sig_t), self._lam(sig_s0)
        h = lam_t - lam_s0
        rks = []
        if order == 2:
            rks.append((self._lam(rk_sigma) - lam_s0) / h)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return rks, torch.stack(R), torch.tensor(b), h_phi_1, B_h

    def step(self, model_output: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        i = self.step_index
        sig = self.sigmas
        m_t = sample - sig[i] * model_output                     # convert_model_output :313-315
        if i > 0 and self.last_sample is not None:               # corrector :690-703
            order = self.this_order
            m0 = self.model_outputs[-1]
            sig_t, sig_s0 = sig[i], sig[i - 1]
            rks, R, b, h_phi_1, B_h = self._coefs(sig_t, sig_s0, order, sig[i - 2] if order == 2 else None)
            alpha_t = 1 - sig_t
            x_t_ = sig_t / sig_s0 * self.last_sample - alpha_t * h_phi_1 * m0
            if order == 1:
                rhos_c = torch.tensor([0.5])
                corr = 0
            else:
                rhos_c = torch.linalg.solve(R, b)
                D1 = (self.model_outputs[-2] - m0) / rks[0]
                corr = rhos_c[0] * D1
            sample = (x_t_ - alpha_t * B_h * (corr + rhos_c[-1] * (m_t - m0))).to(sample.dtype)
        for j in range(self.order - 1):
            self.model_outputs[j] = self.model_outputs[j + 1]
        self.model_outputs[-1] = m_t
        this_order = min(self.order, len(self.timesteps) - i)    # lower_order_final :713-718
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        order = self.this_order                                   # predictor :725-729
        m0 = m_t
        sig_t, sig_s0 = sig[i + 1], sig[i]
        rks, R, b, h_phi_1, B_h = self._coefs(sig_t, sig_s0, order, sig[i - 1] if order == 2 else None)
        alpha_t = 1 - sig_t
        x_t_ = sig_t / sig_s0 * sample - alpha_t * h_phi_1 * m0
        if order == 2:
            D1 = (self.model_outputs[-2] - m0) / rks[0]
            x_t = x_t_ - alpha_t * B_h * (0.5 * D1)
        else:
            x_t = x_t_
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.step_index += 1
        return x_t.to(sample.dtype)


class EulerOracle:
    """EulerScheduler restated -- euler_scheduler.py:5-8 (transform), :35-52, :69-87."""

    def __init__(self, num_train_timesteps=1000):
        self.N = num_train_timesteps

    def set_timesteps(self, num_inference_steps, shift=5.0):
        ts = list(np.linspace(self.N, 1, num_inference_steps, dtype=np.float32)) + [0.0]
        out = []
        for t in ts:
            t = torch.tensor([t]) / self.N
            out.append(shift * t / (1 + (shift - 1) * t) * self.N)
        self.timesteps = torch.tensor(out[:-1])
        return self.timesteps

    def step(self, model_output, timestep, sample):
        idx = int(torch.argmin((self.timesteps - float(timestep)).abs()).item())
        dt_raw = self.timesteps[idx] - self.timesteps[idx + 1] if idx + 1 < len(self.timesteps) else self.timesteps[idx]
        return sample - model_output * (dt_raw.item() / self.N)


class DpmppOracle:
    """FlowDPMSolverMultistepScheduler restated for the configuration generate() builds (any2video.py:524-533):
    order 2, dpmsolver++, midpoint, flow_prediction, lower_order_final, final sigma 0 --
    fm_solvers.py:22-27 (get_sampling_sigmas), :226-291 (set_timesteps), :341-395 (x0 = x - sigma*v),
    :415-468 (first order), :486-557 (second order), :706-797 (step)."""

    def __init__(self, num_train_timesteps=1000):
        self.N = num_train_timesteps

    def set_timesteps(self, sampling_steps: int, shift: float):
        sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
        sigma = shift * sigma / (1 + (shift - 1) * sigma)            # get_sampling_sigmas
        sigmas = 1.0 * sigma / (1 + (1.0 - 1) * sigma)               # set_timesteps(sigmas=..) with config.shift = 1
        timesteps = sigmas * self.N
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)
        self.model_outputs = [None, None]
        self.lower_order_nums = 0
        self.step_index = 0
        return self.timesteps

    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def step(self, model_output, sample):
        i, sig, n = self.step_index, self.sigmas, len(self.timesteps)
        lower_order_final = i == n - 1                                # final_sigmas_type == "zero"
        lower_order_second = (i == n - 2) and n < 15
        m = sample - sig[i] * model_output
        self.model_outputs = [self.model_outputs[1], m]
        sample = sample.to(torch.float32)
        sigma_t, sigma_s0 = sig[i + 1], sig[i]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        if self.lower_order_nums < 1 or lower_order_final:
            prev = (sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * m
        else:
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            h_0 = self._lam(sigma_s0) - self._lam(sig[i - 1])
            r0 = h_0 / h
            D0, D1 = m0, (1.0 / r0) * (m0 - m1)
            prev = ((sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * D0
                    - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * D1)
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.to(model_output.dtype)


class FlowMatchOracle:
    """FlowMatchScheduler (causvid) restated -- basic_flowmatch.py:20-54 as generate() uses it
    (any2video.py:513-517: shift, sigma_min=0, extra_one_step=True, fixed timestep table)."""

    def __init__(self, num_inference_steps, shift, sigma_min=0.0, extra_one_step=True, num_train_timesteps=1000):
        sig = torch.linspace(1.0, sigma_min, num_inference_steps + 1)[:-1] if extra_one_step else \
            torch.linspace(1.0, sigma_min, num_inference_steps)
        self.sigmas = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = self.sigmas * num_train_timesteps

    def step(self, model_output, timestep, sample):
        tid = torch.argmin((self.timesteps - timestep).abs(), dim=0)
        sigma = self.sigmas[tid].reshape(-1, 1, 1, 1)
        sigma_ = 0 if tid + 1 >= len(self.timesteps) else self.sigmas[tid + 1].reshape(-1, 1, 1, 1)
        return sample + model_output * (sigma_ - sigma)


class LcmOracle:
    """LCMScheduler restated -- lcm_scheduler.py:26-76."""

    def __init__(self, num_train_timesteps=1000):
        self.N = num_train_timesteps

    def set_timesteps(self, num_inference_steps, shift):
        n = min(num_inference_steps, 8)
        t = torch.linspace(0, 1, n + 1, dtype=torch.float32)
        sigma_min = 0.003 / 1.002
        sig = sigma_min + (1.0 - sigma_min) * (1 - t)
        self.sigmas = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = self.sigmas[:-1] * self.N
        self.step_index = 0
        return self.timesteps

    def step(self, model_output, sample):
        i = self.step_index
        nxt = self.sigmas[i + 1] if i + 1 < len(self.sigmas) else torch.zeros(())
        self.step_index += 1
        return sample + model_output * (nxt - self.sigmas[i])


def cfg_combine(cond: torch.Tensor, uncond: torch.Tensor, guide_scale: float) -> torch.Tensor:
    """any2video.py:1722: noise_pred = uncond + g * (cond - uncond)."""
    return uncond + guide_scale * (cond - uncond)


def sample_loop(W_hi, cfg: WanConfig, latents: torch.Tensor, ctx: torch.Tensor, ctx_null: torch.Tensor,
                steps: int, shift: float, guide_scale: float, W_lo=None, switch_threshold: float = 0.0,
                guide2_scale: Optional[float] = None, y=None, dtype=torch.bfloat16, exact=False,
                solver: str = "unipc", on_step=None):
    """The t2v/i2v loop body of WanAny2V.generate (any2video.py:1470,1490-1501,
    1626-1634,1702-1722,1733): joint-pass CFG pair, expert switch at t <= switch_threshold
    (:1437-1443), scheduler step on fp32 latents.  Returns (latents, per-step latents)."""
    if solver == "unipc":
        sch = UniPCOracle(); timesteps = sch.set_timesteps(steps, shift)
    else:
        sch = EulerOracle(); timesteps = sch.set_timesteps(steps, shift)
    grid = (latents.shape[2], latents.shape[3] // 2, latents.shape[4] // 2)
    freqs = rope_tables(grid)
    W, g = W_hi, guide_scale
    switched = False
    trace = []
    for t in timesteps:
        if W_lo is not None and not switched and float(t) <= switch_threshold:
            W, switched = W_lo, True
            if guide2_scale is not None:
                g = guide2_scale
        tt = torch.stack([t])
        if g == 1:
            noise = dit_forward([latents], tt, [ctx], W, cfg, y=y, freqs=freqs, dtype=dtype, exact=exact)[0]
        else:
            cond, uncond = dit_forward([latents, latents], tt, [ctx, ctx_null], W, cfg, y=y, freqs=freqs,
                                       dtype=dtype, exact=exact)
            noise = cfg_combine(cond, uncond, g)
        latents = sch.step(noise, latents) if solver == "unipc" else sch.step(noise, t, latents)
        trace.append(latents.clone())
        if on_step is not None:
            on_step(len(trace) - 1, latents)
    return latents, trace


# --------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------
def synth_inputs(cfg: WanConfig, f: int, h: int, w: int, seed: int = 42, text_tokens: int = 77,
                 batch: int = 1):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(batch, cfg.out_dim, f, h, w, generator=g)
    ctx = (torch.randn(batch, cfg.text_len, cfg.text_dim, generator=g) * 0.5)
    ctx[:, text_tokens:] = 0                      # zero padding, any2video.py:590
    ctx_null = (torch.randn(batch, cfg.text_len, cfg.text_dim, generator=g) * 0.5)
    ctx_null[:, 8:] = 0
    y = None
    if cfg.in_dim == 36:
        msk = (torch.rand(4, f, h, w, generator=g) > 0.5).float()
        y = torch.cat([msk, torch.randn(16, f, h, w, generator=g)], dim=0)
    return lat, ctx.to(torch.bfloat16), ctx_null.to(torch.bfloat16), y


def synth_clip_fea(seed: int = 9, images: int = 1):
    """CLIP ViT-H penultimate features as `clip_fea` [images, 257, 1280] bf16 (any2video.py feeds clip.visual output; flf2v: the
    start and the end image, :949-950)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(images, CLIP_TOKENS, CLIP_DIM, generator=g).to(torch.bfloat16)


def synth_vace_context(cfg: WanConfig, f: int, h: int, w: int, seed: int = 13):
    """[vace_in_dim, F, H, W]: the masked-video / mask latents VACE conditions on (any2video.py vace_encode_*)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(cfg.vace_in_dim, f, h, w, generator=g)
