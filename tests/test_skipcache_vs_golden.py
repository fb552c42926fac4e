"""Step-skipping caches (TeaCache / MagCache) on CPU:
  * oracle/skipcache_oracle.py must reproduce tests/golden/skipcache_tiny.npz -- thresholds, per-step should-calc decisions
    and every output of the REFERENCE's own WanModel with `.cache` set (oracle/make_golden_skipcache.py) -- bit-exactly;
  * the product's host logic (wan2gp_amd/skipcache.py: thresholds + decisions) must make the same decisions from the same
    inputs (MagCache needs no tensors at all; TeaCache is fed the oracle's time embeddings)."""
import os

import numpy as np
import pytest
import torch

from oracle import skipcache_oracle as SO
from oracle import wan_oracle as O
from oracle.make_golden_skipcache import MAG_RATIOS, STEPS, TEA_COEF, inputs

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "skipcache_tiny.npz")))
CFG = O.make_config("tiny")


def mk(kind, cls):
    c = cls(cache_type=kind, multiplier=2.0, start_step=1, num_steps=STEPS, skipped_steps=0, previous_residual=None,
            previous_modulated_input=None)
    if kind == "mag":
        c.update({"magcache_thresh": 0, "magcache_K": 2, "def_mag_ratios": list(MAG_RATIOS)})
    else:
        c.update({"coefficients": list(TEA_COEF), "rel_l1_thresh": 0, "accumulated_rel_l1_distance": 0})
    return c


@pytest.fixture(scope="module")
def W():
    return O.synth_weights(CFG, seed=4321)


def test_oracle_magcache_joint_and_single_reproduce_reference(W):
    lats, ts, ctx, ctx_null = inputs(CFG)
    c = mk("mag", SO.Cache)
    c.previous_residual = [None] * 2
    assert SO.magcache_threshold(c, c.start_step, ts, c.multiplier) == pytest.approx(float(G["mag_thresh"][0]), abs=1e-12)
    assert np.array_equal(c.mag_ratios, G["mag_ratios"])
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    with torch.no_grad():
        for i in range(STEPS):
            outs, flags = SO.dit_forward_cached([lats[i], lats[i]], torch.stack([ts[i]]), [ctx, ctx_null], W, CFG, c, 0, i)
            assert [int(f) for f in flags] == G["magj_flags"][i].tolist()
            assert np.array_equal(outs[0].numpy(), G[f"magj_{i}_0"]) and np.array_equal(outs[1].numpy(), G[f"magj_{i}_1"]), i
    assert c.skipped_steps == int(G["magj_skipped"][0])
    c = mk("mag", SO.Cache)
    c.previous_residual = [None] * 2
    SO.magcache_threshold(c, c.start_step, ts, c.multiplier)
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    with torch.no_grad():
        for i in range(STEPS):
            for x_id, cc in enumerate((ctx, ctx_null)):
                outs, flags = SO.dit_forward_cached([lats[i]], torch.stack([ts[i]]), [cc], W, CFG, c, x_id, i)
                assert int(flags[0]) == int(G["mags_flags"][i][x_id])
                assert np.array_equal(outs[0].numpy(), G[f"mags_{i}_{x_id}"]), (i, x_id)


def test_oracle_teacache_reproduces_reference(W):
    lats, ts, ctx, ctx_null = inputs(CFG)
    c = mk("tea", SO.Cache)
    c.previous_residual = [None] * 2
    assert SO.teacache_threshold(c, c.start_step, ts, c.multiplier, W, CFG) == pytest.approx(float(G["tea_thresh"][0]), abs=1e-12)
    with torch.no_grad():
        for i in range(STEPS):
            outs, flags = SO.dit_forward_cached([lats[i], lats[i]], torch.stack([ts[i]]), [ctx, ctx_null], W, CFG, c, 0, i)
            assert int(flags[0]) == int(G["teaj_flags"][i]) and flags[0] == flags[1]
            assert np.array_equal(outs[0].numpy(), G[f"teaj_{i}_0"]) and np.array_equal(outs[1].numpy(), G[f"teaj_{i}_1"]), i
    assert c.skipped_steps == int(G["teaj_skipped"][0]) and 0 < c.skipped_steps < STEPS


def test_product_host_logic_makes_the_reference_decisions(W):
    from wan2gp_amd import skipcache as SK
    _, ts, _, _ = inputs(CFG)
    # MagCache: joint and single passes
    c = mk("mag", SK.SkipStepsCache)
    SK.reset_for_generation(c, 2)
    assert SK.compute_magcache_threshold(c, c.start_step, ts, c.multiplier) == pytest.approx(float(G["mag_thresh"][0]), abs=1e-12)
    assert np.array_equal(c.mag_ratios, G["mag_ratios"])
    flags = [[int(f) for f in SK.decide(c, 2, 0, i)] for i in range(STEPS)]
    assert flags == G["magj_flags"].tolist() and c.skipped_steps == int(G["magj_skipped"][0])
    c = mk("mag", SK.SkipStepsCache)
    SK.reset_for_generation(c, 2)
    SK.compute_magcache_threshold(c, c.start_step, ts, c.multiplier)
    flags = [[int(SK.decide(c, 1, x_id, i)[0]) for x_id in (0, 1)] for i in range(STEPS)]
    assert flags == G["mags_flags"].tolist()
    # TeaCache: decisions from the (oracle-computed, reference-identical) time embeddings
    es = [O.time_embed(torch.stack([t]), W, CFG, torch.bfloat16)[0] for t in ts]
    c = mk("tea", SK.SkipStepsCache)
    SK.reset_for_generation(c, 2)
    assert SK.compute_teacache_threshold(c, c.start_step, es, c.multiplier) == pytest.approx(float(G["tea_thresh"][0]), abs=1e-12)
    flags = []
    for i in range(STEPS):
        f0 = SK.decide(c, 2, 0, i, es[i])
        flags.append(int(f0[0]))
    assert flags == G["teaj_flags"].tolist() and c.skipped_steps == int(G["teaj_skipped"][0])


def test_magcache_ratio_resampling_to_other_step_counts():
    """nearest_interp (model.py:1375-1380): a calibration for 8 steps resampled to 5 and to 12 steps."""
    from wan2gp_amd import skipcache as SK
    for n in (5, 12, 1):
        a, b = mk("mag", SK.SkipStepsCache), mk("mag", SO.Cache)
        ts = list(range(n))
        ta, tb = SK.compute_magcache_threshold(a, 0, ts, 1.5), SO.magcache_threshold(b, 0, ts, 1.5)
        assert ta == tb and np.array_equal(a.mag_ratios, b.mag_ratios) and len(a.mag_ratios) == 2 * n
