"""Randomised comparison of wan2gp_amd/video2video.py with the REFERENCE's own statements (lifted verbatim from
models/wan/any2video.py by oracle/make_golden_v2v.py:build) on a few hundred seeded parameter draws -- step counts, denoising /
masking strengths on and off the rounding edges, keep lists of every alignment, prefix frames, per-frame / single-frame / no masks,
the "U" prompt type that switches the mask off.  Runs where the reference tree is present (the build container); the committed
golden cases of tests/test_v2v_vs_golden.py travel instead.  CPU only."""
import os
import random
import types

import pytest
import torch

from oracle import make_golden_v2v as M
from wan2gp_amd import video2video as V

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(M.REF, "models", "wan")), reason="reference tree not present")


def draw(rng):
    lat = rng.randint(1, 6)
    src = rng.randint(1, lat)
    T = (src - 1) * 4 + 1
    keep_kind = rng.choice(["none", "all", "random", "short", "long"])
    prefix = rng.choice([0, 0, 1, 4, 5])
    n_keep = {"none": 0, "all": max(T - prefix, 1), "random": max(T - prefix, 1), "short": rng.randint(1, 6), "long": T + rng.randint(1, 5)}[keep_kind]
    keep = [True] * n_keep if keep_kind == "all" else [rng.random() < 0.6 for _ in range(n_keep)]
    return dict(src=src, lat=lat, steps=rng.randint(2, 12), ds=rng.choice([0.05, 0.25, 0.3, 0.5, 0.55, 0.6, 0.75, 0.95, 1.0]),
                ms=rng.choice([0.0, 0.1, 0.33, 0.5, 0.99, 1.0]), keep=keep, prefix=prefix, mask=rng.choice([False, True, "one"]),
                vpt=rng.choice(["G", "G", "GU"]))


@pytest.mark.parametrize("block", range(6))
def test_random_draws_reproduce_the_reference_statements(block):
    ns, _ = M.build()
    me = types.SimpleNamespace(device="cpu")
    rng = random.Random(1000 + block)
    seen = set()
    for n in range(40):
        c = draw(rng)
        frames, masks, src, randn, lat, ts = M.inputs(c, 7000 + 100 * block + n)
        s_ref = types.SimpleNamespace(timesteps=ts.clone(), sigmas=torch.cat([ts / 1000, torch.zeros(1)]))
        s_our = types.SimpleNamespace(timesteps=ts.clone(), sigmas=torch.cat([ts / 1000, torch.zeros(1)]))
        st = ns["setup"](me, frames, masks, src, c["lat"], c["steps"], c["ds"], c["ms"], list(c["keep"]), c["prefix"], ts.clone(), s_ref,
                         c["vpt"], False)
        p = V.plan(frames, masks, src, c["lat"], c["steps"], c["ds"], c["ms"], list(c["keep"]), c["prefix"], ts.clone(), s_our,
                   video_prompt_type=c["vpt"])
        assert (p.injection_denoising_step, p.inject_from_start, p.start_step_no, p.masked_steps) == \
            (st["injection_denoising_step"], st["inject_from_start"], st["start_step_no"], st["masked_steps"]), c
        assert list(p.latent_keep_frames) == list(st["latent_keep_frames"]), c
        assert torch.equal(p.timesteps, st["timesteps"]) and torch.equal(s_our.timesteps, s_ref.timesteps) \
            and torch.equal(s_our.sigmas, s_ref.sigmas), c
        assert (p.image_mask_latents is None) == (st["image_mask_latents"] is None), c
        if p.image_mask_latents is not None:
            assert torch.equal(p.image_mask_latents, st["image_mask_latents"]), c
        xr, xo = lat.clone(), lat.clone()
        for i, t in enumerate(st["timesteps"]):
            xr = ns["inject"](xr, randn, src, t, i, c["ds"], st["injection_denoising_step"], st["inject_from_start"], st["latent_keep_frames"])
            xo = V.inject(xo, randn, src, t, i, c["ds"], p)
            assert torch.equal(xr, xo), (c, i)
            xr, xo = xr + 0.1 * torch.roll(xr, 1, dims=-1), xo + 0.1 * torch.roll(xo, 1, dims=-1)
            xr = ns["merge"](xr, randn, src, st["image_mask_latents"], st["timesteps"], i, st["masked_steps"])
            xo = V.merge(xo, randn, src, p.timesteps, i, p)
            assert torch.equal(xr, xo), (c, i)
        seen.add((p.inject_from_start, p.start_step_no > 0, p.image_mask_latents is not None, bool(p.latent_keep_frames)))
    assert len(seen) >= 5, seen                                   # the draws reach the different plans
