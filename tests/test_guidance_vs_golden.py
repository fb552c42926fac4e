"""CFG-Zero* / APG helpers of the sampler loop (wan2gp_amd/guidance.py) against tests/golden/guidance.npz, produced by
oracle/make_golden_guidance.py from the reference's own functions (any2video.py:67-79, multitalk_utils.py:339-381).  Same
torch ops in the same order on the same fp32 inputs: exact equality on CPU."""
import os

import numpy as np
import torch

from wan2gp_amd import guidance as G

GOLD = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "guidance.npz")))
t = lambda k: torch.from_numpy(GOLD[k])


def test_optimized_scale_and_project():
    c, u = t("cond_0"), t("uncond_0")
    assert torch.equal(G.optimized_scale(c.view(1, -1), u.view(1, -1)), t("alpha"))
    par, orth = G.project(u, c)
    assert torch.equal(par, t("proj_par")) and torch.equal(orth, t("proj_orth"))


def test_apg_with_momentum_and_norm_clipping():
    buf = G.MomentumBuffer(-0.75)
    for i in range(4):                                   # step 2 carries a 40x larger difference: the norm clip is active
        c, u = t(f"cond_{i}"), t(f"uncond_{i}")
        assert torch.equal(G.adaptive_projected_guidance(c - u, c, momentum_buffer=buf, norm_threshold=55), t(f"apg_{i}")), i
    c, u = t("cond_1"), t("uncond_1")
    assert torch.equal(G.adaptive_projected_guidance(c - u, c, eta=0.3, norm_threshold=0), t("apg_nomom_eta"))


def test_combine_branches():
    c, u = t("cond_0"), t("uncond_0")
    plain = G.combine(c, u, 4.0, 7)
    assert torch.equal(plain, u + 4.0 * (c - u))
    # CFG-Zero*: goldens are the reference's own branch (any2video.py:1702-1722) executed on these inputs.  For steps
    # <= cfg_zero_step the reference's zeroed prediction is overwritten by the plain CFG line (unscaled uncond).
    assert torch.equal(G.combine(c, u, 4.0, 2, cfg_star_switch=1, cfg_zero_step=5), t("cfgzero_early"))
    assert torch.equal(t("cfgzero_early"), t("cfg_plain")) and torch.equal(plain, G.combine(c, u, 4.0, 9))
    assert torch.equal(G.combine(c, u, 4.0, 9, cfg_star_switch=1, cfg_zero_step=5), t("cfgzero_late"))
    a = t("alpha").view(1, 1, 1, 1)
    assert torch.equal(t("cfgzero_late"), u * a + 4.0 * (c - u * a))
    buf = G.MomentumBuffer(-0.75)
    assert torch.equal(G.combine(c, u, 4.0, 0, apg_switch=1, momentum_buffer=buf), c + 3.0 * t("apg_0"))
