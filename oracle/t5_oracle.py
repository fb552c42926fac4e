"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's UMT5 text encoder
(`models/wan/modules/t5.py`), the step in front of the sampler loop (`any2video.py:587-593`,
SURVEY.md section 8(f) rank 1).  Plain torch on CPU, bf16 plan by default (the reference loads the
encoder with default_dtype=torch.bfloat16, t5.py:689-696); every function cites the lines it follows.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinned by tests/test_t5_oracle_vs_golden.py against tests/golden/t5_small.npz, which
oracle/make_golden_t5.py produced by running the reference's own T5Encoder.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class T5Config:            # umt5_xxl (t5.py:460-472)
    vocab_size: int = 256384
    dim: int = 4096
    dim_attn: int = 4096
    dim_ffn: int = 10240
    num_heads: int = 64
    num_layers: int = 24
    num_buckets: int = 32
    eps: float = 1e-6


SMALL = T5Config(vocab_size=97, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2)


def synth_t5_weights(cfg: T5Config, seed=11, dtype=torch.bfloat16):
    """Random weights with the reference's key names (state dict of T5Encoder) and init scales (t5.py:30-47)."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    W = {"token_embedding.weight": rn(cfg.vocab_size, cfg.dim, std=1.0), "norm.weight": (1 + 0.1 * torch.randn(cfg.dim, generator=g)).to(dtype)}
    for i in range(cfg.num_layers):
        b = f"blocks.{i}."
        W[b + "norm1.weight"] = (1 + 0.1 * torch.randn(cfg.dim, generator=g)).to(dtype)
        W[b + "norm2.weight"] = (1 + 0.1 * torch.randn(cfg.dim, generator=g)).to(dtype)
        W[b + "attn.q.weight"] = rn(cfg.dim_attn, cfg.dim, std=(cfg.dim * cfg.dim_attn) ** -0.5 * 8)
        W[b + "attn.k.weight"] = rn(cfg.dim_attn, cfg.dim, std=cfg.dim ** -0.5)
        W[b + "attn.v.weight"] = rn(cfg.dim_attn, cfg.dim, std=cfg.dim ** -0.5)
        W[b + "attn.o.weight"] = rn(cfg.dim, cfg.dim_attn, std=cfg.dim_attn ** -0.5)
        W[b + "ffn.gate.0.weight"] = rn(cfg.dim_ffn, cfg.dim, std=cfg.dim ** -0.5)
        W[b + "ffn.fc1.weight"] = rn(cfg.dim_ffn, cfg.dim, std=cfg.dim ** -0.5)
        W[b + "ffn.fc2.weight"] = rn(cfg.dim, cfg.dim_ffn, std=cfg.dim_ffn ** -0.5)
        W[b + "pos_embedding.embedding.weight"] = rn(cfg.num_buckets, cfg.num_heads, std=0.5)
    return W


def synth_t5_inputs(cfg: T5Config, B=2, L=40, seed=3):
    """Token ids + padding mask as HuggingfaceTokenizer(..., return_mask=True) returns them (t5.py:711-713)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, cfg.vocab_size, (B, L), generator=g)
    lens = [L - 7 * i for i in range(B)]
    mask = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens):
        mask[i, :n] = 1
        ids[i, n:] = 0
    return ids, mask


def relative_position_bucket(rel_pos, num_buckets=32, max_dist=128):
    """t5.py:244-263, bidirectional branch."""
    nb = num_buckets // 2
    rel_buckets = (rel_pos > 0).long() * nb
    rel_pos = torch.abs(rel_pos)
    max_exact = nb // 2
    rel_pos_large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    rel_pos_large = torch.min(rel_pos_large, torch.full_like(rel_pos_large, nb - 1))
    return rel_buckets + torch.where(rel_pos < max_exact, rel_pos, rel_pos_large)


def relative_bias_table(emb_weight, L, num_buckets=32):
    """[H, 2L-1]: entry (h, r + L - 1) is the bias of relative position r = j - i (t5.py:232-242 evaluated per
    distinct r; pos_bias[h, i, j] = table[h, j - i + L - 1])."""
    r = torch.arange(-(L - 1), L)
    return emb_weight[relative_position_bucket(r, num_buckets)].t().contiguous()


def t5_layer_norm(x, w, eps=1e-6):
    """t5.py:66-71: x * rsqrt(mean(x.float()^2) + eps) is fp32 (bf16 x fp32), cast to the weight dtype, times weight."""
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        y = y.type_as(w)
    return w * y


def gelu(x):
    """t5.py:51-55 (tensor ops in x.dtype)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_attention(x, W, pre, H, pos_bias, mask):
    """t5.py:92-131: no 1/sqrt(d) scaling; scores (x.dtype) + bias, masked with finfo.min, fp32 softmax, cast back."""
    b, L, _ = x.shape
    q = F.linear(x, W[pre + "q.weight"]).view(b, L, H, -1)
    k = F.linear(x, W[pre + "k.weight"]).view(b, L, H, -1)
    v = F.linear(x, W[pre + "v.weight"]).view(b, L, H, -1)
    attn_bias = x.new_zeros(b, H, L, L)
    attn_bias += pos_bias
    if mask is not None:
        attn_bias.masked_fill_(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + attn_bias
    attn = F.softmax(attn.float(), dim=-1).type_as(attn)
    o = torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, L, -1)
    return F.linear(o, W[pre + "o.weight"])


def t5_ffn(x, W, pre):
    """t5.py:148-153."""
    return F.linear(F.linear(x, W[pre + "fc1.weight"]) * gelu(F.linear(x, W[pre + "gate.0.weight"])), W[pre + "fc2.weight"])


def t5_encoder(ids, mask, W, cfg: T5Config):
    """T5Encoder.forward (t5.py:296-306) with per-block relative embeddings (shared_pos=False, t5.py:180-185)."""
    x = W["token_embedding.weight"][ids]
    L = ids.shape[1]
    idx = torch.arange(L).unsqueeze(0) - torch.arange(L).unsqueeze(1) + (L - 1)           # j - i + L - 1
    for i in range(cfg.num_layers):
        b = f"blocks.{i}."
        tab = relative_bias_table(W[b + "pos_embedding.embedding.weight"], L, cfg.num_buckets)   # [H, 2L-1]
        pos_bias = tab[:, idx].unsqueeze(0)                                                      # [1, H, L, L]
        x = x + t5_attention(t5_layer_norm(x, W[b + "norm1.weight"], cfg.eps), W, b + "attn.", cfg.num_heads, pos_bias, mask)
        x = x + t5_ffn(t5_layer_norm(x, W[b + "norm2.weight"], cfg.eps), W, b + "ffn.")
    return t5_layer_norm(x, W["norm.weight"], cfg.eps)
