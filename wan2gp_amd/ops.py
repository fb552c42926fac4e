"""Thin torch-tensor wrappers over the C ABI (one function per libwanhip entry point).

These mirror the reference's operator call sites (names and argument meaning) so the
parity tests read like the reference's code; they hold no arithmetic of their own.
"""
import ctypes
from ctypes import c_float, c_void_p

import torch

from . import lib as _L
from .lib import EPI_GATE_RES, EPI_GELU_TANH, EPI_NONE, EPI_TRANSPOSED, check, ptr, stream_ptr

BF16 = torch.bfloat16


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _L.WanHipError(f"{name} must be a CUDA(HIP) tensor -- libwanhip has no CPU path")
    if t.dtype != dtype:
        raise _L.WanHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _L.WanHipError(f"{name} must be contiguous")


def rmsnorm_rope_(q, k, wq, wk, freqs=None, eps=1e-6, L=None, pos0=0, q_scale=1.0):
    """In-place WanRMSNorm(q)[, WanRMSNorm(k)][, apply_rotary_emb([q,k], freqs)].
    q,k [B,L,d] or [B,L,H,128]; freqs = (cos,sin) [Ltot,128] fp32 or None.
    (model.py:343-350; posemb_layers.py:288-340).  q_scale: fp32 factor folded into q in front of its bf16
    rounding (attention_qscale() for attention(..., q_prescaled=True))."""
    for t, n in ((q, "q"), (k, "k"), (wq, "wq"), (wk, "wk")):
        _req(t, BF16, n)
    d = wq.numel()
    rows = q.numel() // d
    if L is None:
        L = q.shape[1]
    cos = sin = None
    if freqs is not None:
        cos, sin = freqs
        _req(cos, torch.float32, "cos"); _req(sin, torch.float32, "sin")
    check(_L.load().wan_rmsnorm_rope_scaled(ptr(q), ptr(k), ptr(wq), ptr(wk), ptr(cos), ptr(sin), rows, L, pos0, d, eps,
                                            float(q_scale), stream_ptr()), "wan_rmsnorm_rope")
    return q, k


def rmsnorm_rope_pack(x, w, world, heads_per_rank, head_chunks=1, freqs=None, eps=1e-6, L=None, pos0=0, scale=1.0):
    """WanRMSNorm(x)[, RoPE] of one tensor written in the Ulysses exchange's send layout (wan_rmsnorm_rope_pack): x [B,L,d] bf16 (read only)
    -> a flat bf16 tensor of x's size holding [chunk j][world][rows][128 (h0_j+1 - h0_j)], h0_j = j * heads_per_rank // head_chunks."""
    _req(x, BF16, "x"); _req(w, BF16, "w")
    d = w.numel()
    rows = x.numel() // d
    if L is None:
        L = x.shape[1]
    cos = sin = None
    if freqs is not None:
        cos, sin = freqs
        _req(cos, torch.float32, "cos"); _req(sin, torch.float32, "sin")
    out = torch.empty(x.numel(), dtype=BF16, device=x.device)
    check(_L.load().wan_rmsnorm_rope_pack(ptr(x), ptr(out), ptr(w), ptr(cos), ptr(sin), rows, L, pos0, d, eps, float(scale), world, heads_per_rank,
                                          head_chunks, stream_ptr()), "wan_rmsnorm_rope_pack")
    return out


def attention_scratch_words(B, Bk, Lq, H):
    """4-byte words of scratch attention(..., kmax_scratch=) needs: Bk*H maxima + one flag per 256-row workgroup."""
    return int(_L.load().wan_attention_scratch_words(B, Bk, Lq, H))


def attention_qscale():
    """(1/sqrt(128)) * log2(e): the factor attention(..., q_prescaled=True) expects folded into q."""
    return float(_L.load().wan_attention_qscale())


def ln_modulate(x, mod, e, shift_idx, scale_idx, eps=1e-6, out=None):
    """LayerNorm(no affine) then `*= 1+e[scale]; += e[shift]` with e = mod + e0 (model.py:632-638)."""
    _req(x, BF16, "x"); _req(mod, BF16, "mod"); _req(e, BF16, "e")
    d = x.shape[-1]
    n_mod = mod.numel() // d
    rows = x.numel() // d
    nb = e.numel() // (n_mod * d)
    out = torch.empty_like(x) if out is None else out
    check(_L.load().wan_ln_modulate(ptr(x), ptr(out), ptr(mod), ptr(e), n_mod, shift_idx, scale_idx, rows,
                                    rows // nb, d, eps, stream_ptr()), "wan_ln_modulate")
    return out


def ln_affine(x, w, b, eps=1e-6, out=None):
    """WanLayerNorm(elementwise_affine=True) (model.py:199-212, norm3)."""
    for t, n in ((x, "x"), (w, "w"), (b, "b")):
        _req(t, BF16, n)
    d = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    check(_L.load().wan_ln_affine(ptr(x), ptr(out), ptr(w), ptr(b), x.numel() // d, d, eps, stream_ptr()),
          "wan_ln_affine")
    return out


def ln_modulate_amax(x, mod, e, shift_idx, scale_idx, amax_ws, rows_per_slot, eps=1e-6, out=None):
    """ln_modulate that also leaves max |out| of every `rows_per_slot` rows in word 1 of consecutive 64-word fp32 slots of `amax_ws` (float
    bits, atomicMax: zero the words first) -- the abs-max pass of the next fp8 Linear's activation quantisation folded into the producer."""
    _req(x, BF16, "x"); _req(mod, BF16, "mod"); _req(e, BF16, "e"); _req(amax_ws, torch.float32, "amax_ws")
    d = x.shape[-1]
    n_mod = mod.numel() // d
    rows = x.numel() // d
    nb = e.numel() // (n_mod * d)
    if amax_ws.numel() < 64 * ((rows + rows_per_slot - 1) // rows_per_slot):
        raise _L.WanHipError("ln_modulate_amax: amax_ws holds fewer slots than the rows need")
    out = torch.empty_like(x) if out is None else out
    check(_L.load().wan_ln_modulate_amax(ptr(x), ptr(out), ptr(mod), ptr(e), n_mod, shift_idx, scale_idx, rows, rows // nb, d, eps, ptr(amax_ws),
                                         rows_per_slot, stream_ptr()), "wan_ln_modulate_amax")
    return out


def ln_affine_amax(x, w, b, amax_ws, rows_per_slot, eps=1e-6, out=None):
    """ln_affine with the same abs-max side output (see ln_modulate_amax)."""
    for t, n in ((x, "x"), (w, "w"), (b, "b")):
        _req(t, BF16, n)
    _req(amax_ws, torch.float32, "amax_ws")
    d = x.shape[-1]
    rows = x.numel() // d
    if amax_ws.numel() < 64 * ((rows + rows_per_slot - 1) // rows_per_slot):
        raise _L.WanHipError("ln_affine_amax: amax_ws holds fewer slots than the rows need")
    out = torch.empty_like(x) if out is None else out
    check(_L.load().wan_ln_affine_amax(ptr(x), ptr(out), ptr(w), ptr(b), rows, d, eps, ptr(amax_ws), rows_per_slot, stream_ptr()), "wan_ln_affine_amax")
    return out


def fp8_quantize_pre(x, ws, amax_word):
    """The quantising half of fp8_quantize: `ws` (64 fp32 words: a quantisation slot) already holds the tensor's abs-max in ws[amax_word]
    (left there by ln_modulate_amax / ln_affine_amax / linear_fp8_gelu_amax); ws[0] <- scale_a.  Returns the fp8 tensor."""
    _req(x, BF16, "x"); _req(ws, torch.float32, "ws")
    out = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    check(_L.load().wan_fp8_quantize_pre(ptr(x), ptr(out), ptr(ws), x.numel(), int(amax_word), stream_ptr()), "wan_fp8_quantize_pre")
    return out


def linear_fp8_gelu_amax(x_fp8, weight_fp8, weight_scale, bias, amax):
    """linear_fp8(..., epilogue=EPI_GELU_TANH) on an already quantised input (x_fp8 = (fp8 tensor, ws)) that also leaves max |out| in the
    one-float device tensor `amax` (float bits, atomicMax: zero it first)."""
    _req(weight_fp8, torch.float8_e4m3fn, "weight_fp8"); _req(weight_scale, torch.float32, "weight_scale"); _req(bias, BF16, "bias")
    _req(amax, torch.float32, "amax")
    xq, ws = x_fp8
    N, K = weight_fp8.shape
    M = xq.numel() // K
    out = torch.empty(*xq.shape[:-1], N, dtype=BF16, device=xq.device)
    check(_L.load().wan_gemm_fp8_amax(ptr(xq), K, ptr(ws), ptr(weight_fp8), ptr(weight_scale), weight_scale.numel(), ptr(bias), ptr(out), M, N, K,
                                      ptr(amax), stream_ptr()), "wan_gemm_fp8_amax")
    return out


def gated_residual_(x, y, mod=None, e=None, gate_idx=-1):
    """x.addcmul_(y, e[gate]) (model.py:658-660) or x += y when gate_idx < 0."""
    _req(x, BF16, "x"); _req(y, BF16, "y"); _req(mod, BF16, "mod"); _req(e, BF16, "e")
    d = x.shape[-1]
    rows = x.numel() // d
    n_mod = mod.numel() // d if mod is not None else 0
    nb = e.numel() // (n_mod * d) if e is not None else 1
    check(_L.load().wan_gated_residual(ptr(x), ptr(y), ptr(mod), ptr(e), n_mod, gate_idx, rows, rows // nb, d,
                                       stream_ptr()), "wan_gated_residual")
    return x


def nag_combine(x_pos, x_neg, nag_scale, nag_tau, nag_alpha, out=None):
    """Normalized attention guidance on the two text cross-attention results (text_cross_attention, model.py:276-293):
    x_pos / x_neg [..., d] bf16 -> bf16, one rounding per reference statement.  out may be x_pos or x_neg."""
    _req(x_pos, BF16, "x_pos"); _req(x_neg, BF16, "x_neg")
    d = x_pos.shape[-1]
    out = torch.empty_like(x_pos) if out is None else out
    check(_L.load().wan_nag_combine(ptr(x_pos), ptr(x_neg), ptr(out), x_pos.numel() // d, d, float(nag_scale), float(nag_tau),
                                    float(nag_alpha), stream_ptr()), "wan_nag_combine")
    return out


def linear(x, weight, bias=None, epilogue=EPI_NONE, residual=None, mod=None, e=None, gate_idx=-1, out=None,
           ldc=None):
    """nn.Linear on MFMA: bf16(x @ weight.T + bias) with an optional fused epilogue.
    EPI_TRANSPOSED returns the [N, ldc] transposed result (V^T for attention)."""
    _req(x, BF16, "x"); _req(weight, BF16, "weight"); _req(bias, BF16, "bias")
    _req(residual, BF16, "residual"); _req(mod, BF16, "mod"); _req(e, BF16, "e")
    N, K = weight.shape
    M = x.numel() // K
    if epilogue == EPI_TRANSPOSED:
        ldc = ldc or ((M + 63) // 64) * 64
        if out is None:
            out = torch.zeros(N, ldc, dtype=BF16, device=x.device)
    else:
        ldc = N
        if out is None:
            out = torch.empty(*x.shape[:-1], N, dtype=BF16, device=x.device)
    n_mod = mod.numel() // N if mod is not None else 0
    nb = e.numel() // (n_mod * N) if (e is not None and n_mod) else 1
    check(_L.load().wan_gemm_bf16(ptr(x), K, ptr(weight), ptr(bias), ptr(out), ldc, M, N, K, epilogue, ptr(residual),
                                  ptr(mod), ptr(e), n_mod, gate_idx, max(M // nb, 1), stream_ptr()), "wan_gemm_bf16")
    return out


def fp8_quantize(x):
    """_quantize_activation (shared/qtypes/scaled_fp8.py:162-169): x bf16 -> (float8_e4m3fn tensor of x's shape, ws) where
    ws is a 2-float device tensor whose first element is scale_a = absmax / 448."""
    _req(x, BF16, "x")
    out = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    ws = torch.empty(2, dtype=torch.float32, device=x.device)
    check(_L.load().wan_fp8_quantize(ptr(x), ptr(out), ptr(ws), x.numel(), stream_ptr()), "wan_fp8_quantize")
    return out, ws


def linear_fp8(x, weight_fp8, weight_scale, bias=None, epilogue=EPI_NONE, residual=None, mod=None, e=None, gate_idx=-1,
               out=None, ldc=None, x_fp8=None):
    """ScaledFP8WeightTensor._linear_scaled (scaled_fp8.py:324-380) with an optional fused epilogue: x [..., K] bf16 is
    quantised per tensor (or pass x_fp8 = fp8_quantize(x) to share it between Linears of the same input), weight_fp8 [N, K]
    float8_e4m3fn, weight_scale fp32 scalar or [N] / [N, 1]."""
    _req(weight_fp8, torch.float8_e4m3fn, "weight_fp8"); _req(weight_scale, torch.float32, "weight_scale")
    _req(bias, BF16, "bias"); _req(residual, BF16, "residual"); _req(mod, BF16, "mod"); _req(e, BF16, "e")
    N, K = weight_fp8.shape
    if weight_scale.numel() not in (1, N):
        raise _L.WanHipError(f"weight_scale must have 1 or {N} elements")
    xq, ws = fp8_quantize(x) if x_fp8 is None else x_fp8
    M = xq.numel() // K
    if epilogue == EPI_TRANSPOSED:
        ldc = ldc or ((M + 63) // 64) * 64
        if out is None:
            out = torch.zeros(N, ldc, dtype=BF16, device=xq.device)
    else:
        ldc = N
        if out is None:
            out = torch.empty(*xq.shape[:-1], N, dtype=BF16, device=xq.device)
    n_mod = mod.numel() // N if mod is not None else 0
    nb = e.numel() // (n_mod * N) if (e is not None and n_mod) else 1
    check(_L.load().wan_gemm_fp8(ptr(xq), K, ptr(ws), ptr(weight_fp8), ptr(weight_scale), weight_scale.numel(), ptr(bias), ptr(out),
                                 ldc, M, N, K, epilogue, ptr(residual), ptr(mod), ptr(e), n_mod, gate_idx, max(M // nb, 1),
                                 stream_ptr()), "wan_gemm_fp8")
    return out


def permute16(src, A, B):
    """Block transpose src [A][B][blk] -> [B][A][blk] (wan_permute16: the head-group-major re-packs around the Ulysses all-to-alls);
    blk = the rest of the tensor, a multiple of 16 bytes.  Returns a new contiguous tensor of src's dtype."""
    if not src.is_cuda or not src.is_contiguous():
        raise _L.WanHipError("permute16: a contiguous CUDA tensor is required")
    nbytes = src.numel() * src.element_size()
    if A * B == 0 or nbytes % (A * B):
        raise _L.WanHipError(f"permute16: {nbytes} bytes do not split into {A} x {B} blocks")
    out = torch.empty_like(src)
    check(_L.load().wan_permute16(ptr(src), ptr(out), A, B, nbytes // (A * B), stream_ptr()), "wan_permute16")
    return out


def permute16_ex(src, dst, A, B, nbytes, src_a_pitch, src_b_pitch, dst_a_pitch, dst_b_pitch):
    """dst[b * dst_b_pitch + a * dst_a_pitch + i] = src[a * src_a_pitch + b * src_b_pitch + i], a < A, b < B, i < nbytes (all in BYTES,
    multiples of 16; wan_permute16_ex: the per-head-chunk re-packs of the chunked Ulysses exchange).  src / dst: CUDA tensors whose
    data pointers are the bases (pass slices for offsets); the caller guarantees the extents."""
    if not (src.is_cuda and dst.is_cuda):
        raise _L.WanHipError("permute16_ex: CUDA tensors are required")
    check(_L.load().wan_permute16_ex(ptr(src), ptr(dst), A, B, nbytes, src_a_pitch, src_b_pitch, dst_a_pitch, dst_b_pitch, stream_ptr()),
          "wan_permute16_ex")
    return dst


def transpose_v(v, ldv=None):
    """[B,L,H,128] (or [B,L,C]) -> V^T [B, C, ldv] with zero padding."""
    _req(v, BF16, "v")
    B, L = v.shape[0], v.shape[1]
    C = v.numel() // (B * L)
    ldv = ldv or ((L + 63) // 64) * 64
    vt = torch.empty(B, C, ldv, dtype=BF16, device=v.device)
    check(_L.load().wan_transpose_v(ptr(v), ptr(vt), B, L, ldv, C, stream_ptr()), "wan_transpose_v")
    return vt


def attention(q, k, vt, Lk=None, out=None, nseg=1, k_seg_stride=0, vt_seg_stride=0, Bk=None, q_prescaled=False,
              kmax_scratch=None):
    """softmax(q k^T / sqrt(128)) v with v given transposed.  q [B,Lq,H,128], k [Bk,Lk,H,128], vt [Bk,H*128,ldv].
    nseg > 1: k / vt hold `nseg` gathered segments ([seg][Bk][Lk][H*128], [seg][Bk][H*128][ldv]); pass Lk, Bk
    and the segment strides (elements) explicitly.  q_prescaled: q already holds q * attention_qscale().
    kmax_scratch: fp32 device scratch of attention_scratch_words(B, Bk, Lq, H) elements for the K pre-pass
    (wan_attention_bounded); False = no pre-pass (the kernel's lazy-max loop); None = the library's own scratch."""
    _req(q, BF16, "q"); _req(k, BF16, "k"); _req(vt, BF16, "vt")
    B, Lq, H, D = q.shape
    if D != 128:
        raise _L.WanHipError("head_dim must be 128")
    if Bk is None:
        Bk = k.shape[0]
    Lk = Lk if Lk is not None else k.shape[1]
    ldv = vt.shape[-1]
    out = torch.empty_like(q) if out is None else out
    if kmax_scratch is not None:
        if kmax_scratch is False:
            km = None
        else:
            _req(kmax_scratch, torch.float32, "kmax_scratch")
            need = attention_scratch_words(B, Bk, Lq, H)
            if kmax_scratch.numel() < need:
                raise _L.WanHipError(f"kmax_scratch needs {need} 4-byte words")
            km = kmax_scratch
        check(_L.load().wan_attention_bounded(ptr(q), ptr(k), ptr(vt), ptr(out), B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride,
                                              vt_seg_stride, 1 if q_prescaled else 0, ptr(km), stream_ptr()), "wan_attention_bounded")
        return out
    fn = _L.load().wan_attention_prescaled if q_prescaled else _L.load().wan_attention_seg
    check(fn(ptr(q), ptr(k), ptr(vt), ptr(out), B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, stream_ptr()),
          "wan_attention")
    return out


def attention_sp(q, k_local, vt_local, k_all, vt_all, own_seg, out=None):
    """Sequence-parallel self-attention, local segment first (wan_attention_sp_local + wan_attention_sp_remote): q [B,Lq,H,128]
    pre-scaled, k_local [B,Lk,H,128] / vt_local [B,H*128,ldv] = this rank's segment, k_all [nseg,B,Lk,H,128] / vt_all
    [nseg,B,H*128,ldv] = the gathered segments (own_seg is not read from them).  Returns (out, between) where `between()` is
    where a caller waits for its all-gathers; this helper runs both phases back to back."""
    for t, n in ((q, "q"), (k_local, "k_local"), (vt_local, "vt_local"), (k_all, "k_all"), (vt_all, "vt_all")):
        _req(t, BF16, n)
    B, Lq, H, _ = q.shape
    nseg, Lk, ldv = k_all.shape[0], k_all.shape[2], vt_all.shape[-1]
    lib = _L.load()
    scratch = torch.zeros(int(lib.wan_attention_scratch_words(B, B, Lq, H)), dtype=torch.float32, device=q.device)
    raw = torch.empty(int(lib.wan_attention_raw_words(B, Lq, H)), dtype=torch.float32, device=q.device)
    out = torch.empty_like(q) if out is None else out
    check(lib.wan_attention_sp_local(ptr(q), ptr(k_local), ptr(vt_local), B, Lq, Lk, ldv, H, ptr(scratch), ptr(raw), stream_ptr()),
          "wan_attention_sp_local")
    check(lib.wan_attention_sp_remote(ptr(q), ptr(k_all), ptr(vt_all), ptr(out), B, Lq, Lk, ldv, H, nseg, B * Lk * H * 128,
                                      B * H * 128 * ldv, own_seg, ptr(scratch), ptr(raw), stream_ptr()), "wan_attention_sp_remote")
    return out, scratch


def pay_attention(qkv_list, dropout_p=0., softmax_scale=None, causal=False, window_size=(-1, -1),
                  deterministic=False, version=None, force_attention=None, attention_mask=None, recycle_q=False,
                  q_lens=None, k_lens=None):
    """Drop-in for shared/attention.py:360-373 `pay_attention` (attention mode "hip").
    Same conventions: q,k,v [B,L,H,D] with D contiguous, the list is consumed (:403), k/v
    batch 1 is broadcast (:415-416), the result has q's dtype (:563), recycle_q lets the
    output reuse q's storage.  Unsupported options raise instead of silently approximating."""
    if causal or attention_mask is not None or dropout_p != 0. or q_lens is not None or k_lens is not None \
            or window_size != (-1, -1):
        raise _L.WanHipError("pay_attention[hip]: only unmasked non-causal attention is implemented")
    q, k, v = qkv_list
    qkv_list.clear()
    if softmax_scale is not None and abs(softmax_scale - q.shape[-1] ** -0.5) > 1e-9:
        raise _L.WanHipError("pay_attention[hip]: only the default 1/sqrt(D) scale is implemented")
    out_dtype = q.dtype
    q = q.to(BF16).contiguous(); k = k.to(BF16).contiguous(); v = v.to(BF16).contiguous()
    vt = transpose_v(v)
    o = attention(q, k, vt, out=q if (recycle_q and out_dtype == BF16) else None)
    return o.type(out_dtype)


def t5_attention(q, k, v, relbias, mask=None, out=None):
    """T5Attention core (models/wan/modules/t5.py:109-131), head_dim 64, no scaling.  q,k,v [B,L,H*64] bf16;
    relbias [H, 2L-1] bf16 (relbias[h][j-i+L-1]); mask [B,L] int32 (0 = padding key) or None."""
    _req(q, BF16, "q"); _req(k, BF16, "k"); _req(v, BF16, "v"); _req(relbias, BF16, "relbias")
    B, L, C = q.shape
    H = C // 64
    if mask is not None:
        _req(mask, torch.int32, "mask")
    out = torch.empty_like(q) if out is None else out
    check(_L.load().wan_t5_attention(ptr(q), ptr(k), ptr(v), ptr(relbias), ptr(mask), ptr(out), B, L, H, stream_ptr()),
          "wan_t5_attention")
    return out


def mul(a, b, out=None):
    """bf16(a * b) elementwise (T5FeedForward gate product, t5.py:149)."""
    _req(a, BF16, "a"); _req(b, BF16, "b")
    out = torch.empty_like(a) if out is None else out
    check(_L.load().wan_mul_bf16(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr()), "wan_mul_bf16")
    return out


def lora_accumulate(acc, lora_B, lora_A, scale):
    """acc[N,K] (fp32) += scale * lora_B[N,r] @ lora_A[r,K] (fp32 operands)."""
    F32 = torch.float32
    _req(acc, F32, "acc"); _req(lora_B, F32, "lora_B"); _req(lora_A, F32, "lora_A")
    N, K = acc.shape
    r = lora_A.shape[0]
    if tuple(lora_B.shape) != (N, r) or tuple(lora_A.shape) != (r, K):
        raise _L.WanHipError(f"lora_accumulate: B {list(lora_B.shape)} @ A {list(lora_A.shape)} does not give {[N, K]}")
    check(_L.load().wan_lora_accumulate(ptr(acc), ptr(lora_B), ptr(lora_A), float(scale), N, K, r, stream_ptr()), "wan_lora_accumulate")
    return acc


def axpy_f32(acc, x, alpha):
    _req(acc, torch.float32, "acc"); _req(x, torch.float32, "x")
    if acc.numel() != x.numel():
        raise _L.WanHipError("axpy_f32: size mismatch")
    check(_L.load().wan_axpy_f32(ptr(acc), ptr(x), float(alpha), acc.numel(), stream_ptr()), "wan_axpy_f32")
    return acc


def add_f32_into_bf16_(w, acc):
    """w = bf16(float(w) + acc) in place: the single rounding of a LoRA merge."""
    _req(w, BF16, "w"); _req(acc, torch.float32, "acc")
    if acc.numel() != w.numel():
        raise _L.WanHipError("add_f32_into_bf16_: size mismatch")
    check(_L.load().wan_add_f32_into_bf16(ptr(w), ptr(acc), w.numel(), stream_ptr()), "wan_add_f32_into_bf16")
    return w


def dequant_i8(data, scale):
    """bf16(float(data[n,k]) * scale[n]): optimum-quanto qint8 weight -> resident bf16."""
    _req(data, torch.int8, "data"); _req(scale, torch.float32, "scale")
    N, K = data.shape
    if scale.numel() != N:
        raise _L.WanHipError("dequant_i8: one scale per output row expected")
    out = torch.empty(N, K, dtype=BF16, device=data.device)
    check(_L.load().wan_dequant_i8(ptr(data), ptr(scale), ptr(out), N, K, stream_ptr()), "wan_dequant_i8")
    return out


def patch_embed(x, w, bias, y=None):
    """patch_embedding Conv3d(k=s=(1,2,2)) fp32 -> bf16 tokens [B,L,d] (model.py:1631,1731)."""
    _req(x, torch.float32, "x"); _req(w, torch.float32, "w"); _req(bias, torch.float32, "bias")
    _req(y, torch.float32, "y")
    B, Cin, F, H, W = x.shape
    d = w.shape[0]
    Cy = 0 if y is None else y.shape[0]
    out = torch.empty(B, F * (H // 2) * (W // 2), d, dtype=BF16, device=x.device)
    check(_L.load().wan_patch_embed(ptr(x), ptr(y), ptr(w), ptr(bias), ptr(out), B, Cin, Cy, F, H, W, d, stream_ptr()),
          "wan_patch_embed")
    return out


def head(x, hmod, e, w, bias, grid, eps=1e-6):
    """Head.forward + unpatchify -> fp32 [B,16,F,H,W] (model.py:847-865,2100-2126)."""
    _req(x, BF16, "x"); _req(hmod, torch.float32, "hmod"); _req(e, BF16, "e")
    _req(w, torch.float32, "w"); _req(bias, torch.float32, "bias")
    B, L, d = x.shape
    F, Hg, Wg = grid
    tmp = torch.empty_like(x)
    nout = w.shape[0]                                                      # 4 * out_dim: 64, or 192 for the ti2v 5B model
    out = torch.empty(B, nout // 4, F, Hg * 2, Wg * 2, dtype=torch.float32, device=x.device)
    check(_L.load().wan_head_n(ptr(x), ptr(hmod), ptr(e), ptr(w), ptr(bias), ptr(tmp), ptr(out), B, F, Hg, Wg, d, eps, nout,
                               stream_ptr()), "wan_head")
    return out


def unpatchify(tok, grid):
    _req(tok, torch.float32, "tok")
    B, nout = tok.shape[0], tok.shape[-1]
    F, Hg, Wg = grid
    out = torch.empty(B, nout // 4, F, Hg * 2, Wg * 2, dtype=torch.float32, device=tok.device)
    check(_L.load().wan_unpatchify_n(ptr(tok), ptr(out), B, F, Hg, Wg, nout, stream_ptr()), "wan_unpatchify")
    return out


def lincomb(tensors, coefs, out=None):
    """out = sum_i coefs[i] * tensors[i] on fp32 latents (scheduler / CFG arithmetic)."""
    n = len(tensors)
    for i, t in enumerate(tensors):
        _req(t, torch.float32, f"in[{i}]")
    out = torch.empty_like(tensors[0]) if out is None else out
    P = (c_void_p * n)(*[t.data_ptr() for t in tensors])
    C = (c_float * n)(*[float(c) for c in coefs])
    check(_L.load().wan_lincomb(ptr(out), n, P, C, tensors[0].numel(), stream_ptr()), "wan_lincomb")
    return out


def gemv(x, weight, bias):
    _req(x, BF16, "x"); _req(weight, BF16, "weight"); _req(bias, BF16, "bias")
    N, K = weight.shape
    M = x.numel() // K
    out = torch.empty(M, N, dtype=BF16, device=x.device)
    check(_L.load().wan_gemv_bf16(ptr(x), ptr(weight), ptr(bias), ptr(out), M, N, K, stream_ptr()), "wan_gemv_bf16")
    return out


def sinusoid(t, dim):
    """sinusoidal_embedding_1d(dim, t) -> bf16 [n, dim] (model.py:32-42)."""
    _req(t, torch.float32, "t")
    out = torch.empty(t.numel(), dim, dtype=BF16, device=t.device)
    check(_L.load().wan_sinusoid(ptr(t), ptr(out), t.numel(), dim, stream_ptr()), "wan_sinusoid")
    return out


def silu(x):
    _req(x, BF16, "x")
    out = torch.empty_like(x)
    check(_L.load().wan_act_bf16(ptr(x), ptr(out), x.numel(), 1, stream_ptr()), "wan_act_bf16")
    return out


def cfg_combine(cond, uncond, guide_scale, out=None):
    """noise_pred = uncond + g * (cond - uncond)   (any2video.py:1722), fp32."""
    _req(cond, torch.float32, "cond"); _req(uncond, torch.float32, "uncond")
    out = torch.empty_like(cond) if out is None else out
    check(_L.load().wan_cfg_combine(ptr(out), ptr(cond), ptr(uncond), float(guide_scale), cond.numel(), stream_ptr()),
          "wan_cfg_combine")
    return out
