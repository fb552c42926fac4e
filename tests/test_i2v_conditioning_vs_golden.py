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


GE = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "i2v_cond_end.npz")))


def _end_cases():
    from oracle.make_golden_i2v_cond import end_cases
    return list(enumerate(end_cases()))


@pytest.mark.parametrize("i,c", _end_cases(), ids=lambda v: v["name"] if isinstance(v, dict) else str(v))
def test_i2v_conditioning_with_an_end_image_reproduces_reference(i, c):
    """Start + end image (any2video.py:684-704, :747-751, :768-769): Wan2.2 i2v keeps the frame count and closes the clip with
    the end image; the Wan2.1 i2v model (`add`) appends one frame / latent frame encoded with any_end_frame."""
    pipe = WanAny2VHIP(model=None, vae=FakeVAE(), device="cpu")
    v, e = make_video(c["P"], 60 + i), make_video(1, 80 + i)
    y, ext = pipe.build_i2v_conditioning(v if c["P"] > 1 else v[:, 0], c["frames"], 32, 48, 0, c["amp"], image_end=e[:, 0],
                                         add_frames_for_end_image=c["add"])
    assert torch.equal(y, torch.from_numpy(GE[c["name"] + "_y"]))
    assert torch.equal(ext, torch.from_numpy(GE[c["name"] + "_ext"]))
    assert y.shape[1] == (c["frames"] - 1) // 4 + 1 + (1 if c["add"] else 0)
