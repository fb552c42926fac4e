"""Video-to-video inside the sampler loop (`wan2gp_amd/video2video.py`: denoising strength, kept frames, masked regeneration)
against tests/golden/v2v.npz -- the reference's own statements of `WanAny2V.generate` (any2video.py:1007-1042, :1504-1515,
:1737-1740, lifted verbatim by oracle/make_golden_v2v.py) on the same seeded inputs.  Exact equality on CPU."""
import os
import types

import numpy as np
import pytest
import torch

from oracle.make_golden_v2v import cases, inputs
from wan2gp_amd import video2video as V

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "v2v.npz")))


@pytest.mark.parametrize("n,c", list(enumerate(cases())), ids=lambda v: v["name"] if isinstance(v, dict) else str(v))
def test_v2v_plan_inject_merge_reproduce_reference(n, c):
    frames, masks, src, randn, lat, ts = inputs(c, 90 + n)
    sched = types.SimpleNamespace(timesteps=ts.clone(), sigmas=torch.cat([ts / 1000, torch.zeros(1)]))
    p = V.plan(frames, masks, src, c["lat"], c["steps"], c["ds"], c["ms"], list(c["keep"]), c["prefix"], ts.clone(), sched)
    k = c["name"] + "_"
    assert [p.injection_denoising_step, int(p.inject_from_start), p.start_step_no, p.masked_steps] == G[k + "ints"].tolist()
    assert [int(v) for v in p.latent_keep_frames] == G[k + "keep"].tolist()
    assert np.array_equal(p.timesteps.numpy(), G[k + "timesteps"])
    assert np.array_equal(sched.timesteps.numpy(), G[k + "sched_timesteps"]) and np.array_equal(sched.sigmas.numpy(), G[k + "sched_sigmas"])
    if p.image_mask_latents is None:
        assert k + "mask_latents" not in G
    else:
        assert np.array_equal(p.image_mask_latents.numpy(), G[k + "mask_latents"])
    x = lat.clone()
    for i, t in enumerate(p.timesteps):
        x = V.inject(x, randn, src, t, i, c["ds"], p)
        assert torch.equal(x, torch.from_numpy(G[k + f"inj_{i}"])), i
        x = x + 0.1 * torch.roll(x, 1, dims=-1)                       # the same stand-in for the model + scheduler step
        x = V.merge(x, randn, src, p.timesteps, i, p)
        assert torch.equal(x, torch.from_numpy(G[k + f"mrg_{i}"])), i


def test_cases_cover_both_plans_and_the_mask_forms():
    ints = {c["name"]: G[c["name"] + "_ints"].tolist() for c in cases()}
    assert ints["full_clip"][1] == 0 and ints["full_clip"][2] > 0          # schedule cut short: start_step_no > 0
    assert ints["short_source"][1] == 1 and ints["keep_list"][1] == 1      # re-injection from the start
    assert 0 in G["keep_list_keep"].tolist() and 1 in G["keep_list_keep"].tolist()
    assert G["one_frame_mask_mask_latents"].shape[2] == 1 and G["short_source_mask_latents"].shape[2] == 3
    assert not V.plan(torch.zeros(3, 5, 8, 8), torch.ones(1, 5, 8, 8), torch.zeros(1, 16, 2, 1, 1), 2, 4, 1.0, 1.0, [], 0,
                      torch.arange(4.), None, video_prompt_type="GU").masked_steps  # "U": the mask is not used
