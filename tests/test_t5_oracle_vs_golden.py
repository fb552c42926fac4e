"""Pins oracle/t5_oracle.py (CPU restatement of models/wan/modules/t5.py) to tests/golden/t5_small.npz, which
oracle/make_golden_t5.py produced by running the reference's own T5Encoder / T5RelativeEmbedding / T5LayerNorm /
GELU on the seeded synthetic weights and inputs.  Bit-exact in both plans (same torch ops in the same order)."""
import os

import numpy as np
import pytest
import torch

from oracle import t5_oracle as T

G = os.path.join(os.path.dirname(__file__), "golden", "t5_small.npz")


@pytest.mark.parametrize("tag,dtype", [("bf16", torch.bfloat16), ("fp32", torch.float32)])
def test_t5_oracle_reproduces_reference(tag, dtype):
    g = dict(np.load(G))
    cfg = T.SMALL
    W = T.synth_t5_weights(cfg, dtype=dtype)
    ids, mask = T.synth_t5_inputs(cfg)
    assert list(ids.shape) == list(g["shape"])
    L = ids.shape[1]
    # relative position bias: table lookup == the reference's [1, H, L, L] tensor
    tab = T.relative_bias_table(W["blocks.0.pos_embedding.embedding.weight"], L, cfg.num_buckets)
    idx = torch.arange(L).unsqueeze(0) - torch.arange(L).unsqueeze(1) + (L - 1)
    assert np.array_equal(tab[:, idx].unsqueeze(0).float().numpy(), g[f"posbias_{tag}"])
    x0 = W["token_embedding.weight"][ids]
    assert np.array_equal(T.t5_layer_norm(x0, W["blocks.0.norm1.weight"], cfg.eps).float().numpy(), g[f"ln_{tag}"])
    assert np.array_equal(T.gelu(x0).float().numpy(), g[f"gelu_{tag}"])
    with torch.no_grad():
        y = T.t5_encoder(ids, mask, W, cfg)
    assert y.dtype == dtype
    assert np.array_equal(y.float().numpy(), g[f"out_{tag}"])


def test_t5_padding_does_not_leak():
    """Tokens behind the mask must not influence the valid positions (masked_fill with finfo.min, t5.py:119-123)."""
    cfg = T.SMALL
    W = T.synth_t5_weights(cfg)
    ids, mask = T.synth_t5_inputs(cfg)
    ids2 = ids.clone()
    ids2[1, int(mask[1].sum()):] = 5
    with torch.no_grad():
        a = T.t5_encoder(ids, mask, W, cfg); b = T.t5_encoder(ids2, mask, W, cfg)
    n = int(mask[1].sum())
    assert torch.equal(a[1, :n], b[1, :n]) and torch.equal(a[0], b[0])
