"""i2v conditioning (`WanAny2VHIP.build_i2v_conditioning`: mask + VAE latents of the known frames, prefix-video continuation,
motion amplitude) against tests/golden/i2v_cond.npz -- the reference's own statements of `WanAny2V.generate`
(any2video.py:699-782, lifted verbatim by oracle/make_golden_i2v_cond.py) run with the same deterministic stand-in VAE.
Exact equality on CPU."""
import os

import numpy as np
import pytest
import torch

from oracle.make_golden_i2v_cond import FakeVAE, cases, make_video
from wan2gp_amd.pipeline import WanAny2VHIP

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "i2v_cond.npz")))


@pytest.mark.parametrize("i,c", list(enumerate(cases())), ids=lambda v: v["name"] if isinstance(v, dict) else str(v))
def test_i2v_conditioning_reproduces_reference(i, c):
    pipe = WanAny2VHIP(model=None, vae=FakeVAE(), device="cpu")
    v = make_video(c["P"], 40 + i)
    y, ext = pipe.build_i2v_conditioning(v if c["P"] > 1 else v[:, 0], c["frames"], 32, 48, 0, c["amp"])
    assert torch.equal(y, torch.from_numpy(G[c["name"] + "_y"]))
    assert torch.equal(ext, torch.from_numpy(G[c["name"] + "_ext"]))
