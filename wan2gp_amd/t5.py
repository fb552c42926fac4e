"""T5EncoderHIP -- the UMT5-XXL text encoder of `models/wan/modules/t5.py` on the HIP kernels.

Mirrors `T5Encoder` (t5.py:266-306) / `T5EncoderModel.__call__` (t5.py:709-716): token embedding lookup, 24 blocks of
[T5LayerNorm -> T5Attention with a per-block bidirectional relative-position bias (shared_pos=False) -> residual,
T5LayerNorm -> gated-GELU feed-forward -> residual], final T5LayerNorm.  Weights are resident bf16 (the reference loads
the encoder with default_dtype bf16, t5.py:689-696); state-dict keys are the reference's.  Linear layers run on
wan_gemm_bf16 (residual adds fused as the GATE_RES epilogue, GELU fused into the gate projection), T5LayerNorm on the
RMSNorm kernel, attention on wan_t5_attention.  Tokenisation (HuggingfaceTokenizer, tokenizers.py) stays with the
caller: pass token ids and the padding mask.
"""
import torch

from . import ops
from .lib import WanHipError

BF16 = torch.bfloat16


def relative_position_bucket(rel_pos, num_buckets=32, max_dist=128):
    """Bidirectional bucket of relative position j - i (t5.py:244-263)."""
    import math
    nb = num_buckets // 2
    rel_buckets = (rel_pos > 0).long() * nb
    rel_pos = torch.abs(rel_pos)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return rel_buckets + torch.where(rel_pos < max_exact, rel_pos, large)


class T5EncoderHIP:
    def __init__(self, vocab_size=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24,
                 num_buckets=32, eps=1e-6, device="cuda"):
        if dim_attn != num_heads * 64:
            raise WanHipError("T5EncoderHIP: head_dim must be 64 (umt5-xxl: 4096 / 64 heads)")
        self.vocab_size, self.dim, self.dim_attn, self.dim_ffn = vocab_size, dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.eps = num_heads, num_layers, num_buckets, eps
        self.device = torch.device(device)
        self.w = None
        self._bias_cache = {}

    def load_state_dict(self, sd):
        need = ["token_embedding.weight", "norm.weight"]
        for i in range(self.num_layers):
            b = f"blocks.{i}."
            need += [b + n for n in ("norm1.weight", "norm2.weight", "attn.q.weight", "attn.k.weight", "attn.v.weight",
                                     "attn.o.weight", "ffn.gate.0.weight", "ffn.fc1.weight", "ffn.fc2.weight",
                                     "pos_embedding.embedding.weight")]
        missing = [k for k in need if k not in sd]
        if missing:
            raise WanHipError(f"T5EncoderHIP.load_state_dict: missing keys {missing[:4]}...")
        self.w = {k: sd[k].to(device=self.device, dtype=BF16).contiguous() for k in need}
        self._bias_cache.clear()
        return self

    def _relbias(self, i, L):
        """[H, 2L-1] bf16: bias of relative position r = j - i at column r + L - 1 (t5.py:232-242 per distinct r)."""
        key = (i, L)
        if key not in self._bias_cache:
            r = torch.arange(-(L - 1), L, device=self.device)
            emb = self.w[f"blocks.{i}.pos_embedding.embedding.weight"]
            self._bias_cache[key] = emb[relative_position_bucket(r, self.num_buckets)].t().contiguous()
        return self._bias_cache[key]

    def _norm(self, x, w):
        y = x.clone()
        ops.rmsnorm_rope_(y, None, w, None, None, eps=self.eps)
        return y

    def forward(self, ids, mask=None):
        """ids [B, L] int64, mask [B, L] (0 = padding) -> [B, L, dim] bf16 (T5Encoder.forward, t5.py:296-306)."""
        if self.w is None:
            raise WanHipError("T5EncoderHIP: load_state_dict first")
        w = self.w
        ids = ids.to(self.device)
        B, L = ids.shape
        m32 = None if mask is None else mask.to(device=self.device, dtype=torch.int32).contiguous()
        x = w["token_embedding.weight"][ids].contiguous()                      # [B, L, dim]
        for i in range(self.num_layers):
            b = f"blocks.{i}."
            h = self._norm(x, w[b + "norm1.weight"])
            q = ops.linear(h, w[b + "attn.q.weight"]); k = ops.linear(h, w[b + "attn.k.weight"]); v = ops.linear(h, w[b + "attn.v.weight"])
            a = ops.t5_attention(q, k, v, self._relbias(i, L), m32)
            x = ops.linear(a, w[b + "attn.o.weight"], epilogue=ops.EPI_GATE_RES, residual=x, out=x)          # x + attn(...)
            h = self._norm(x, w[b + "norm2.weight"])
            g = ops.linear(h, w[b + "ffn.gate.0.weight"], epilogue=ops.EPI_GELU_TANH)                          # GELU(gate(x))
            f = ops.linear(h, w[b + "ffn.fc1.weight"])
            x = ops.linear(ops.mul(f, g, out=f), w[b + "ffn.fc2.weight"], epilogue=ops.EPI_GATE_RES, residual=x, out=x)
        return self._norm(x, w["norm.weight"])

    __call__ = forward

    def encode(self, ids, mask):
        """T5EncoderModel.__call__ after tokenisation (t5.py:709-716): one [seq_len_i, dim] tensor per prompt."""
        ctx = self.forward(ids, mask)
        lens = mask.gt(0).sum(dim=1).long().tolist()
        return [u[:n] for u, n in zip(ctx, lens)]


class T5EncoderModelHIP:
    """Drop-in for `T5EncoderModel` (t5.py:676-716) behind `WanPipeline(text_encoder=...)`: `tokenizer` is the
    reference's `HuggingfaceTokenizer(name, seq_len=text_len, clean='whitespace')` (tokenizers.py:44-82; host-side
    string work, out of scope) or any callable `(texts, return_mask=True, add_special_tokens=True) -> (ids, mask)`."""

    def __init__(self, text_len, tokenizer, state_dict=None, device="cuda", **encoder_kw):
        self.text_len, self.tokenizer = text_len, tokenizer
        self.model = T5EncoderHIP(device=device, **encoder_kw)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)

    def __call__(self, texts, device=None):
        ids, mask = self.tokenizer(texts, return_mask=True, add_special_tokens=True)
        return self.model.encode(torch.as_tensor(ids), torch.as_tensor(mask))
