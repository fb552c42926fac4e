"""Op-level wrappers of the mixed-precision transformer plan's kernels (csrc/mixed_ops.hip, include/wanhip.h `wan_mx_*`):
`mixed_precision_transformer` of the reference (wgp.py:4039 -> any2video.py:190 -> model.py:1330-1371) keeps the residual stream, e / e0
and every modulate / gated residual in fp32 between bf16 Linears.  Used by the -m gpu parity tests and by WanModelHIP.time_embedding (TeaCache's fp32 `e`); the forward
driver (wan_dit_forward, which switches to this plan when the registered time_projection / norm3 weights are fp32: csrc/dit.hip
`ctx_is_mixed`) calls the same C entries."""
import torch

from . import lib as _L
from .lib import check, ptr, stream_ptr

BF16, F32 = torch.bfloat16, torch.float32


def _req(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise _L.WanHipError(f"{name}: expected a contiguous CUDA tensor of {dtype}, got "
                             f"{getattr(t, 'dtype', type(t))} cuda={getattr(t, 'is_cuda', None)}")


def ln_modulate(x, mod, e0, shift_idx, scale_idx, eps=1e-6):
    """x fp32 [..., d], mod bf16 [1, n_mod, d], e0 fp32 [batches, n_mod, d] -> bf16 (model.py:634-638 in the mixed plan)."""
    _req(x, F32, "x"); _req(mod, BF16, "mod"); _req(e0, F32, "e0")
    d = x.shape[-1]
    n_mod = mod.numel() // d
    rows = x.numel() // d
    nb = e0.numel() // (n_mod * d)
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(_L.load().wan_mx_ln_modulate(ptr(x), ptr(out), ptr(mod), ptr(e0), n_mod, shift_idx, scale_idx, rows, rows // nb, d, eps, stream_ptr()),
          "wan_mx_ln_modulate")
    return out


def ln_affine(x, w, b, eps=1e-6):
    """norm3 with fp32 weight / bias on the fp32 stream -> bf16 (model.py:664-665)."""
    _req(x, F32, "x"); _req(w, F32, "w"); _req(b, F32, "b")
    d = x.shape[-1]
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(_L.load().wan_mx_ln_affine(ptr(x), ptr(out), ptr(w), ptr(b), x.numel() // d, d, eps, stream_ptr()), "wan_mx_ln_affine")
    return out


def gated_residual_(x, y, mod=None, e0=None, gate_idx=-1):
    """x (fp32, in place) += y (bf16) * (mod[gate_idx] + e0[b][gate_idx]); gate_idx < 0: x += y."""
    _req(x, F32, "x"); _req(y, BF16, "y")
    d = x.shape[-1]
    rows = x.numel() // d
    n_mod, nb = 1, 1
    if gate_idx >= 0:
        _req(mod, BF16, "mod"); _req(e0, F32, "e0")
        n_mod = mod.numel() // d
        nb = e0.numel() // (n_mod * d)
    check(_L.load().wan_mx_gated_residual(ptr(x), ptr(y), ptr(mod) if gate_idx >= 0 else None, ptr(e0) if gate_idx >= 0 else None, n_mod, gate_idx,
                                          rows, rows // nb, d, stream_ptr()), "wan_mx_gated_residual")
    return x


def linear_gated_residual_(x, a, weight, bias, mod=None, e0=None, gate_idx=-1):
    """x (fp32 [M, N], in place) += bf16(a @ weight.T + bias) * (mod[gate_idx] + e0[b][gate_idx]) in ONE launch (wan_gemm_bf16_res32: the update
    in the tile GEMM's epilogue; shapes that do not fit it run the Linear + wan_mx_gated_residual pair -- bit-identical either way)."""
    _req(x, F32, "x"); _req(a, BF16, "a"); _req(weight, BF16, "weight"); _req(bias, BF16, "bias")
    N, K = weight.shape
    M = a.numel() // K
    n_mod, rpb = 1, M
    if gate_idx >= 0:
        _req(mod, BF16, "mod"); _req(e0, F32, "e0")
        n_mod = mod.numel() // N
        rpb = M // (e0.numel() // (n_mod * N))
    tmp = torch.empty(M, N, dtype=BF16, device=x.device)
    check(_L.load().wan_gemm_bf16_res32(ptr(a), K, ptr(weight), ptr(bias), ptr(x), ptr(tmp), M, N, K, ptr(mod) if gate_idx >= 0 else None,
                                        ptr(e0) if gate_idx >= 0 else None, n_mod, gate_idx, rpb, stream_ptr()), "wan_gemm_bf16_res32")
    return x


def patch_embed(x, w, bias, y=None):
    """x fp32 [Cin, F, H, W] (+ y fp32 [Cy, F, H, W]) -> fp32 [1, L, d] (model.py:1620-1631 with an fp32 modulation dtype)."""
    _req(x, F32, "x"); _req(w, F32, "w"); _req(bias, F32, "bias")
    if y is not None:
        _req(y, F32, "y")
    Cin, F, H, W = x.shape
    d = w.shape[0]
    L = F * (H // 2) * (W // 2)
    out = torch.empty(1, L, d, dtype=F32, device=x.device)
    check(_L.load().wan_mx_patch_embed(ptr(x), ptr(y) if y is not None else None, ptr(w), ptr(bias), ptr(out), Cin, 0 if y is None else y.shape[0],
                                       F, H, W, d, 0, L, stream_ptr()), "wan_mx_patch_embed")
    return out


def sinusoid(t, dim, device="cuda"):
    out = torch.empty(1, dim, dtype=F32, device=device)
    check(_L.load().wan_mx_sinusoid(float(t), ptr(out), dim, stream_ptr()), "wan_mx_sinusoid")
    return out


def linear_f32(a, w, bias, silu_input=False):
    """fp32 Linear on few rows (time MLP / projection under the fp32 lock): act(a) @ w.T + bias."""
    _req(a, F32, "a"); _req(w, F32, "w"); _req(bias, F32, "bias")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=F32, device=a.device)
    check(_L.load().wan_mx_linear_f32(ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K, 1 if silu_input else 0, stream_ptr()), "wan_mx_linear_f32")
    return out


def head(x, hmod, e, w, bias, eps=1e-6):
    """Head.forward on the fp32 stream: x [1, L, d] fp32, hmod [1, 2, d] fp32, e [nb, d] fp32 -> [1, L, nout] fp32 (token-major)."""
    _req(x, F32, "x"); _req(hmod, F32, "hmod"); _req(e, F32, "e"); _req(w, F32, "w"); _req(bias, F32, "bias")
    d = x.shape[-1]
    L = x.numel() // d
    nout = w.shape[0]
    tmp = torch.empty(L, d, dtype=F32, device=x.device)
    out = torch.empty(1, L, nout, dtype=F32, device=x.device)
    check(_L.load().wan_mx_head(ptr(x), ptr(hmod), ptr(e), ptr(w), ptr(bias), ptr(tmp), ptr(out), L, d, eps, L // e.shape[0], nout, stream_ptr()),
          "wan_mx_head")
    return out
