"""Pins the oracle's normalized-attention-guidance branch (oracle/wan_oracle.py: nag_combine, cross_attention(nag=...)) to
tests/golden/nag.npz -- outputs of the REFERENCE's own WanModel / WanT2VCrossAttention with offload.shared_state["_nag_*"]
set the way WanAny2V.generate sets it (oracle/make_golden_nag.py; models/wan/modules/model.py:245-293,
models/wan/any2video.py:607-608).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "nag.npz")))
BF = torch.bfloat16


def cross_inputs(cfg, L=96, seed=31):                 # = oracle/make_golden_nag.py:cross_inputs
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, L, cfg.dim, generator=g).to(BF)
    ctx = (0.5 * torch.randn(2, cfg.text_len, cfg.dim, generator=g)).to(BF)
    ctx[0, 77:] = 0
    ctx[1, 8:] = 0
    return x, ctx


@pytest.mark.parametrize("tag", ["strong", "mild"])
def test_cross_attention_nag_bit_exact(tag):
    cfg = O.make_config("tiny")
    W = O.synth_weights(cfg)
    x, ctx = cross_inputs(cfg)
    nag = tuple(float(v) for v in G[f"cross_{tag}_nag"])
    o = O.cross_attention(x, ctx, W, "blocks.1.cross_attn.", cfg, False, nag)
    assert torch.equal(o.float(), torch.from_numpy(G[f"cross_{tag}"]))
    clipped, rows = (int(v) for v in G[f"cross_{tag}_clipped_rows"])
    assert (0 < clipped < rows) if tag == "strong" else clipped == 0      # the fixture exercises both arms of the norm clip


def test_nag_off_is_the_plain_path():
    """nag_scale <= 1 (or a batch-1 context) never enters the branch (model.py:260)."""
    cfg = O.make_config("tiny")
    W = O.synth_weights(cfg)
    x, ctx = cross_inputs(cfg)
    plain = O.cross_attention(x, ctx[:1], W, "blocks.1.cross_attn.", cfg, False)
    assert torch.equal(plain, O.cross_attention(x, ctx[:1], W, "blocks.1.cross_attn.", cfg, False, (11.0, 2.5, 0.25)))
    assert not torch.equal(plain, O.cross_attention(x, ctx, W, "blocks.1.cross_attn.", cfg, False, (11.0, 2.5, 0.25)))


@pytest.mark.parametrize("name", ["small", "tiny_i2v21"])
def test_forward_with_nag_bit_exact(name):
    cfg = O.make_config(name)
    W = O.synth_weights(cfg)
    f, h, w = (int(v) for v in G[f"fwd_{name}_shape"])
    lat, c, cn, y = O.synth_inputs(cfg, f, h, w)
    clip = O.synth_clip_fea() if cfg.model_type == "i2v" else None
    t = torch.tensor([int(G[f"fwd_{name}_t"][0])], dtype=torch.int64)
    nag = tuple(float(v) for v in G["fwd_nag"])
    cond, uncond = O.dit_forward([lat, lat], t, [torch.cat([c, cn]), cn], W, cfg, y=y, clip_fea=clip, nag=nag)
    assert torch.equal(cond, torch.from_numpy(G[f"fwd_{name}_cond"]))
    assert torch.equal(uncond, torch.from_numpy(G[f"fwd_{name}_uncond"]))
    plain = O.dit_forward([lat], t, [c], W, cfg, y=y, clip_fea=clip)[0]
    assert not torch.equal(plain, cond)


def test_skip_layer_guidance_bit_exact():
    """perturbation_layers (any2video.py:1502, model.py:2025-2028): the listed blocks run for the first stream of the joint pass
    only -- the uncond stream passes through them unchanged.  Oracle only: the HIP forward does not take the argument yet
    (WanModelHIP.forward raises / delegates), DESIGN.md section 8."""
    cfg = O.make_config("small")
    W = O.synth_weights(cfg)
    lat, c, cn, _ = O.synth_inputs(cfg, 3, 10, 14)
    t = torch.tensor([412], dtype=torch.int64)
    cond, uncond = O.dit_forward([lat, lat], t, [c, cn], W, cfg, perturbation_layers=[1])
    assert torch.equal(cond, torch.from_numpy(G["slg_small_cond"])) and torch.equal(uncond, torch.from_numpy(G["slg_small_uncond"]))
    plain = O.dit_forward([lat, lat], t, [c, cn], W, cfg)
    assert torch.equal(plain[0], cond) and not torch.equal(plain[1], uncond)
