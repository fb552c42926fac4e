"""CPU: oracle/fp8_oracle.py against tests/golden/fp8_linear.npz (the reference's own scaled-fp8 functions executed on CPU,
oracle/make_golden_fp8.py): activation quantisation, the scaled fp8 x fp8 Linear and the dequantised fallback, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import fp8_oracle as F

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "fp8_linear.npz")))
CASES = sorted({k.split("/")[0] for k in G})
SHAPES = {"per_tensor_bias": (70, 256), "per_row_bias": (70, 256), "per_row_col_nobias": (33, 128), "batched_3d_per_row": (2, 40, 192),
          "zero_input": (16, 64)}


def bf(a):
    return torch.from_numpy(a.copy()).view(torch.bfloat16)


def load(name):
    x = bf(G[name + "/x"]).reshape(SHAPES[name])
    w = torch.from_numpy(G[name + "/w"].copy()).view(torch.float8_e4m3fn)
    scale = torch.from_numpy(G[name + "/scale"].copy())
    bias = bf(G[name + "/bias"]) if name + "/bias" in G else None
    return x, w, scale, bias


@pytest.mark.parametrize("name", CASES)
def test_fp8_oracle_is_bit_exact_to_the_reference(name):
    x, w, scale, bias = load(name)
    xq, sa = F.quantize_activation(x.reshape(-1, x.shape[-1]))
    assert np.array_equal(xq.view(torch.uint8).numpy(), G[name + "/x_fp8"])
    assert float(sa) == float(G[name + "/scale_a"])
    got = F.linear_scaled(x, w, scale, bias)
    assert got.dtype == torch.bfloat16
    assert np.array_equal(got.view(torch.int16).numpy(), G[name + "/out_scaled"]), \
        (got.float() - bf(G[name + "/out_scaled"]).float().reshape(got.shape)).abs().max()
    assert np.array_equal(F.dequantize(w, scale).view(torch.int16).numpy(), G[name + "/dequant"])
    fb = F.linear_fallback(x, w, scale, bias)
    assert np.array_equal(fb.view(torch.int16).numpy(), G[name + "/out_fallback"])


def test_fp8_plans_differ_by_activation_quantisation_only():
    """The scaled plan is ~2.7e-2 (relative, Frobenius) from the dequantised one on unit-variance data: the e4m3 rounding of
    the activations (3 mantissa bits).  This sets the scale for any bf16-vs-fp8 comparison of a whole model."""
    x, w, scale, bias = load("per_row_bias")
    a, b = F.linear_scaled(x, w, scale, bias).float(), F.linear_fallback(x, w, scale, bias).float()
    r = ((a - b).norm() / b.norm()).item()
    assert 1e-2 < r < 5e-2, r


def test_quantize_weight_round_trip():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(64, 128, generator=g) * 0.02
    q, s = F.quantize_weight(w)
    assert q.dtype == torch.float8_e4m3fn and s.shape == (64,)
    assert ((F.dequantize(q, s, torch.float32) - w).abs() <= w.abs().amax(dim=1, keepdim=True) * 2 ** -4 + 1e-9).all()


@pytest.mark.parametrize("tag,per_row", [("row", True), ("tensor", False)])
def test_oracle_fp8_forward_is_bit_exact_to_the_reference_model(tag, per_row):
    """tests/golden/forward_tiny_fp8.npz: the reference's WanModel with its block Linears running the reference's
    `_linear_scaled` on fp8 weights (oracle/make_golden_fp8.py forward).  The oracle's dit_forward on the same fp8 checkpoint
    must reproduce it bit for bit -- the fp8 Linear inside the pinned block structure."""
    from oracle import wan_oracle as O
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "forward_tiny_fp8.npz")))
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config("tiny")
    W8 = O.quantize_checkpoint_fp8(O.synth_weights(cfg), per_row=per_row)
    assert sum(v.dtype == torch.float8_e4m3fn for v in W8.values()) == 10 * cfg.num_layers
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w)
    out = O.dit_forward([lat, lat], torch.tensor([int(g["t"][0])]), [ctx, ctx_null], W8, cfg, dtype=torch.bfloat16)
    assert np.array_equal(out[0].numpy(), g[f"cond_{tag}"]) and np.array_equal(out[1].numpy(), g[f"uncond_{tag}"])
