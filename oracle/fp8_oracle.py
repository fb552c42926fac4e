"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's scaled-fp8 Linear (shared/qtypes/scaled_fp8.py), the
arithmetic of `*_fp8` checkpoints (BASELINE.json configs[4]: "fp8 MFMA weights").

Checkpoint layout (QLinearScaledFP8._load_from_state_dict, scaled_fp8.py:563-637): `<name>.weight` float8_e4m3fn [N, K] (OCP
e4m3fn, max 448), `<name>.scale_weight` (or `.weight_scale`) fp32 -- a scalar or one value per output row ([N] or [N, 1]) --
and `<name>.bias` in the model dtype.

Two execution plans exist in the reference (QLinearScaledFP8.forward, :546-561):
  * linear_scaled    (:324-380; taken on a GPU whose torch._scaled_mm supports fp8 -- MI300/MI355 do): the ACTIVATION is
                     quantised too, dynamically and per tensor (:162-169: absmax / 448, the quotient x / scale evaluated in
                     the activation dtype, clamp, cast), the product runs fp8 x fp8 with fp32 accumulation and is scaled by
                     scale_a * scale_b; a per-row weight scale is applied afterwards on the 16-bit result, then the bias.
                     This is the plan libwanhip implements on the fp8 MFMA (wan_gemm_fp8).
  * linear_fallback  (:306-322, and `dequantize` :294-304): weights dequantised to the model dtype (w.to(bf16) * scale.to(bf16)),
                     ordinary bf16 Linear.  What the reference runs on CPU; restated here as the accuracy anchor of the plan above.

Pinned by oracle/make_golden_fp8.py -> tests/golden/fp8_linear.npz: the reference's own functions, lifted with `ast` and
executed on CPU (torch._scaled_mm has a CPU implementation); tests/test_fp8_oracle_vs_golden.py requires bit equality.
"""
import torch

FP8 = torch.float8_e4m3fn
FP8_MAX = float(torch.finfo(FP8).max)          # 448.0


def quantize_activation(x: torch.Tensor):
    """scaled_fp8.py:162-169 `_quantize_activation`: per-tensor dynamic scale.  Returns (x_fp8, scale_a fp32 scalar)."""
    absmax = x.abs().max().float()
    scale = absmax / FP8_MAX
    scale = torch.where(absmax > 0, scale, torch.ones_like(scale))
    scale_act = scale.to(dtype=x.dtype)                       # the divisor is rounded to the activation dtype ...
    q = (x / scale_act).clamp(-FP8_MAX, FP8_MAX).to(FP8)      # ... and so is the quotient, before the fp8 cast (RNE)
    return q, scale.reshape(()).to(torch.float32)


def weight_scale_kind(scale: torch.Tensor, weight: torch.Tensor):
    """scaled_fp8.py:150-159 `_scaled_mm_weight_scale`: (per-tensor scale_b | 1, per-row output scale | None)."""
    if scale.numel() == 1:
        return scale.reshape(()), None
    if scale.ndim == 1 and scale.shape[0] == weight.shape[0]:
        return torch.ones((), dtype=torch.float32), scale
    if scale.ndim == 2 and scale.shape[0] == weight.shape[0] and scale.shape[1] == 1:
        return torch.ones((), dtype=torch.float32), scale.reshape(weight.shape[0])
    raise ValueError("unsupported fp8 weight scale shape %s" % (tuple(scale.shape),))


def linear_scaled(x: torch.Tensor, w_fp8: torch.Tensor, scale: torch.Tensor, bias=None):
    """scaled_fp8.py:324-380 `_linear_scaled`.  x [..., K] bf16; w_fp8 [N, K] float8_e4m3fn; returns [..., N] in x.dtype.
    The fp8 x fp8 product with fp32 accumulation is exact up to summation order (every product of two e4m3 values is exact
    in fp32), so `out = fp32_sum * scale_a * scale_b (+ bias)` rounded once to x.dtype is what torch._scaled_mm returns."""
    scale_b, output_scale = weight_scale_kind(scale, w_fp8)
    x2d = x.reshape(-1, x.shape[-1])
    x_fp8, scale_a = quantize_activation(x2d)
    acc = x_fp8.float() @ w_fp8.float().t()
    out = acc * scale_a * scale_b.to(torch.float32)
    if output_scale is None:
        if bias is not None:
            out = out + bias.to(x.dtype).float()               # _scaled_mm adds the bias in fp32, in front of the output rounding
        out = out.to(x.dtype)
    else:
        out = out.to(x.dtype)
        out = out * output_scale.to(dtype=out.dtype).view(1, -1)      # `out *= output_scale` on the 16-bit tensor: one rounding
        if bias is not None:
            out = out + bias.to(out.dtype).view(1, -1)                  # `out += bias`: another
    return out.reshape(*x.shape[:-1], w_fp8.shape[0])


def dequantize(w_fp8: torch.Tensor, scale: torch.Tensor, dtype=torch.bfloat16):
    """scaled_fp8.py:294-304: data.to(dtype) * scale.to(dtype) (per tensor or per row)."""
    out = w_fp8.to(dtype)
    s = scale.to(dtype)
    if s.numel() == 1:
        return out * s
    return out * s.reshape(w_fp8.shape[0], 1)


def linear_fallback(x: torch.Tensor, w_fp8: torch.Tensor, scale: torch.Tensor, bias=None, dtype=torch.bfloat16):
    """scaled_fp8.py:306-322 `_linear_fallback`: dequantised weights, plain Linear in the model dtype."""
    w = dequantize(w_fp8, scale, dtype)
    out = torch.matmul(x.to(dtype).reshape(-1, x.shape[-1]), w.t()).reshape(*x.shape[:-1], w.shape[0])
    if bias is not None:
        out = out + bias
    return out


def quantize_weight(w: torch.Tensor, per_row: bool = True, fp8_dtype=None):
    """Synthetic fp8 checkpoint tensors from a bf16/fp32 weight (the reference ships no quantiser for this format -- the
    files come pre-quantised): absmax scaling per output row (or per tensor) onto the e4m3 range."""
    w = w.float()
    fp8_dtype = FP8 if fp8_dtype is None else fp8_dtype                # (float8_e5m2: the `scaled_float8_e5m2` qtype, scaled_fp8.py:17,34-49)
    fmax = float(torch.finfo(fp8_dtype).max)
    amax = w.abs().amax(dim=1, keepdim=True) if per_row else w.abs().max().reshape(1, 1)
    scale = (amax / fmax).clamp_min(1e-12)
    q = (w / scale).clamp(-fmax, fmax).to(fp8_dtype)
    return q, (scale.reshape(-1) if per_row else scale.reshape(()))
