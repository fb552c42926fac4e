"""-m gpu: TeaCache / MagCache on the resident HIP model against
tests/golden/skipcache_tiny.npz -- the REFERENCE's own WanModel with `.cache` set, 8 steps, three scenarios
(oracle/make_golden_skipcache.py).  Decisions must equal the reference's exactly; outputs within the forward tolerance of
tests/test_gpu_model.py (relative L2 <= 2.5e-2 against the reference's bf16 result) on computed AND skipped steps."""
import os

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O
from oracle.make_golden_skipcache import MAG_RATIOS, STEPS, TEA_COEF, inputs

pytestmark = pytest.mark.gpu
G_BF = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "skipcache_tiny.npz")))
# the same scenarios under the reference's `mixed_precision_transformer` locks (oracle/make_golden_skipcache.py mixed): fp32 residual
# stream, fp32 TeaCache `e`, fp32 previous_residual -- the combination the round-4 advisor found crashing mid-generation
G_MX = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "skipcache_tiny_mixed.npz")))
G = G_BF
CFG = O.make_config("tiny")


def rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module", params=["bf16", "mixed"])
def model(request):
    global G
    from wan2gp_amd.model import WanModelHIP
    mixed = request.param == "mixed"
    G = G_MX if mixed else G_BF
    m = WanModelHIP(model_type=CFG.model_type, dim=CFG.dim, ffn_dim=CFG.ffn_dim, num_heads=CFG.num_heads, num_layers=CFG.num_layers,
                    in_dim=CFG.in_dim, mixed_precision=mixed)
    return m.load_state_dict(O.synth_weights(CFG, seed=4321, mixed=mixed))


def mk(kind):
    from wan2gp_amd.skipcache import SkipStepsCache, reset_for_generation
    c = SkipStepsCache(cache_type=kind, multiplier=2.0, start_step=1, num_steps=STEPS, skipped_steps=0, previous_residual=None,
                       previous_modulated_input=None)
    if kind == "mag":
        c.update({"magcache_thresh": 0, "magcache_K": 2, "def_mag_ratios": list(MAG_RATIOS)})
    else:
        c.update({"coefficients": list(TEA_COEF), "rel_l1_thresh": 0, "accumulated_rel_l1_distance": 0})
    reset_for_generation(c, 2)
    return c


def test_magcache_joint_pass_vs_reference(model):
    lats, ts, ctx, ctx_null = inputs(CFG)
    c = model.cache = mk("mag")
    try:
        assert model.compute_magcache_threshold(c.start_step, ts, c.multiplier) == pytest.approx(float(G["mag_thresh"][0]), abs=1e-12)
        worst = 0.0
        for i in range(STEPS):
            outs = model([lats[i].cuda(), lats[i].cuda()], t=torch.stack([ts[i]]), context=[ctx.cuda(), ctx_null.cuda()], real_step_no=i,
                         current_step_no=i)
            assert [int(s == 0) for s in c.accumulated_steps] == G["magj_flags"][i].tolist(), i
            for k in range(2):
                worst = max(worst, rel(outs[k], G[f"magj_{i}_{k}"]))
        print(f"MagCache joint: skipped {c.skipped_steps}/{STEPS}, worst rel err {worst:.4f}")
        assert c.skipped_steps == int(G["magj_skipped"][0]) and worst <= 2.5e-2
    finally:
        model.cache = None


def test_magcache_single_passes_vs_reference(model):
    lats, ts, ctx, ctx_null = inputs(CFG)
    c = model.cache = mk("mag")
    try:
        model.compute_magcache_threshold(c.start_step, ts, c.multiplier)
        worst = 0.0
        for i in range(STEPS):
            for x_id, cc in enumerate((ctx, ctx_null)):
                out = model([lats[i].cuda()], t=torch.stack([ts[i]]), context=[cc.cuda()], real_step_no=i, current_step_no=i, x_id=x_id)[0]
                assert int(c.accumulated_steps[x_id] == 0) == int(G["mags_flags"][i][x_id])
                worst = max(worst, rel(out, G[f"mags_{i}_{x_id}"]))
        assert worst <= 2.5e-2, worst
    finally:
        model.cache = None


def test_teacache_joint_pass_vs_reference(model):
    lats, ts, ctx, ctx_null = inputs(CFG)
    c = model.cache = mk("tea")
    try:
        assert model.compute_teacache_threshold(c.start_step, ts, c.multiplier) == pytest.approx(float(G["tea_thresh"][0]), abs=1e-12)
        worst, flags = 0.0, []
        for i in range(STEPS):
            outs = model([lats[i].cuda(), lats[i].cuda()], t=torch.stack([ts[i]]), context=[ctx.cuda(), ctx_null.cuda()], real_step_no=i,
                         current_step_no=i)
            flags.append(int(c.should_calc))
            for k in range(2):
                worst = max(worst, rel(outs[k], G[f"teaj_{i}_{k}"]))
        print(f"TeaCache joint: flags {flags}, worst rel err {worst:.4f}")
        assert flags == G["teaj_flags"].tolist() and c.skipped_steps == int(G["teaj_skipped"][0]) and worst <= 2.5e-2
    finally:
        model.cache = None


def test_teacache_with_per_frame_timesteps(model):
    """Per-frame timesteps (t of shape [1, F]: model.py:1812-1818) together with TeaCache -- refused until round 6.  The reference's decision
    reads e = time_embedding(sinusoidal(t.flatten())): F rows, the relative L1 a mean over all of them (model.py:1954).  (a) F equal
    timesteps: decisions, skip count and outputs of the scalar-t run (the rows are copies of its one row); (b) frame 0 held at t = 0 (the
    ti2v image-conditioning pattern, any2video.py:1496-1499): decisions equal a host-side restatement on the stacked embeddings."""
    from wan2gp_amd import skipcache
    lats, ts, ctx, ctx_null = inputs(CFG)
    F = lats[0].shape[-3]
    assert F > 1
    runs = {}
    for name in ("scalar", "equal_frames", "frame0_clean"):
        c = model.cache = mk("tea")
        try:
            model.compute_teacache_threshold(c.start_step, ts, c.multiplier)
            flags, outs_all, prev, acc, want = [], [], None, 0.0, []
            for i in range(STEPS):
                tf = ts[i].reshape(1).repeat(F)
                if name == "frame0_clean":
                    tf[0] = 0
                t = torch.stack([ts[i]]) if name == "scalar" else tf.reshape(1, F)
                if name == "frame0_clean":                                                     # model.py:1947-1962 restated on the F-row embedding
                    e = torch.cat([model.time_embedding(float(v)) for v in tf], 0)
                    if i <= c.start_step or i == c.num_steps - 1 or prev is None:
                        w, acc = True, 0.0
                    else:
                        acc += abs(np.poly1d(c.coefficients)(skipcache._rel_l1(e, prev)))
                        w = not (acc < c.rel_l1_thresh)
                        acc = 0.0 if w else acc
                    prev = e
                    want.append(int(w))
                outs = model([lats[i].cuda(), lats[i].cuda()], t=t, context=[ctx.cuda(), ctx_null.cuda()], real_step_no=i, current_step_no=i)
                flags.append(int(c.should_calc))
                outs_all.append([o.cpu() for o in outs])
            runs[name] = (flags, c.skipped_steps, outs_all)
            if name == "frame0_clean":
                assert flags == want, (flags, want)
                assert all(torch.isfinite(o).all() for step in outs_all for o in step)
        finally:
            model.cache = None
    assert runs["equal_frames"][0] == runs["scalar"][0] == G["teaj_flags"].tolist() and runs["equal_frames"][1] == runs["scalar"][1]
    worst = max(rel(a, b) for sa, sb in zip(runs["equal_frames"][2], runs["scalar"][2]) for a, b in zip(sa, sb))
    print(f"TeaCache + per-frame t: flags {runs['equal_frames'][0]} (scalar-t run's), per-frame vs scalar outputs {worst:.2e}; frame-0-clean flags {runs['frame0_clean'][0]}")
    assert worst <= 1e-2


@pytest.mark.parametrize("kind", ["mag", "tea"])
def test_cache_on_the_unconditional_rank_of_cfg_parallelism(model, kind):
    """A step-skipping cache together with CFG parallelism (refused until round 6): the rank that runs only the unconditional stream
    (`cfg_parallel_stream` 1, set by sp.CfgParallel.attach) makes the conditional stream's decision itself before its own pass.  One GPU plays
    that rank -- x_id-1 passes only, a cache of its own -- against the single-process order (x_id 0 then 1 per step): the same decisions at
    every step and the same outputs (MagCache also against the reference's own single passes in the golden file)."""
    lats, ts, ctx, ctx_null = inputs(CFG)

    def setup():
        c = model.cache = mk(kind)
        (model.compute_magcache_threshold if kind == "mag" else model.compute_teacache_threshold)(c.start_step, ts, c.multiplier)
        return c
    try:
        c = setup()
        want_flags, want = [], []
        for i in range(STEPS):
            for x_id, cc in enumerate((ctx, ctx_null)):
                out = model([lats[i].cuda()], t=torch.stack([ts[i]]), context=[cc.cuda()], real_step_no=i, current_step_no=i, x_id=x_id)[0]
                if x_id == 1:
                    want.append(out.cpu())
                    want_flags.append(int(c.accumulated_steps[1] == 0) if kind == "mag" else int(c.should_calc))
        skipped = c.skipped_steps
        c = setup()
        model.cfg_parallel_stream = 1
        got_flags, worst = [], 0.0
        for i in range(STEPS):
            out = model([lats[i].cuda()], t=torch.stack([ts[i]]), context=[ctx_null.cuda()], real_step_no=i, current_step_no=i, x_id=1)[0]
            got_flags.append(int(c.accumulated_steps[1] == 0) if kind == "mag" else int(c.should_calc))
            assert torch.equal(out.cpu(), want[i]), (kind, i)
            if kind == "mag":
                worst = max(worst, rel(out, G[f"mags_{i}_1"]))
        assert got_flags == want_flags and c.skipped_steps == skipped and 0 in got_flags[2:] and worst <= 2.5e-2, (got_flags, want_flags, worst)
    finally:
        model.cache = None
        model.cfg_parallel_stream = None


def test_skip_reapplies_the_stored_residual(model):
    """Same inputs computed, then skipped: patch_embed(x) + (x_after - x_before) must give the computed output back up to
    the two bf16 roundings of the residual round trip; a skipped stream without a stored residual is an error."""
    from wan2gp_amd.lib import WanHipError
    from wan2gp_amd.skipcache import SkipStepsCache
    lats, ts, ctx, ctx_null = inputs(CFG)
    t = torch.stack([ts[3]])
    plain = model([lats[3].cuda()], t=t, context=[ctx.cuda()])[0]
    c = model.cache = SkipStepsCache(cache_type="mag", start_step=0, one_for_all=False, magcache_K=4, magcache_thresh=10.0, skipped_steps=0,
                                     mag_ratios=np.ones(64), accumulated_err=[0.0, 0.0], accumulated_steps=[0, 0],
                                     accumulated_ratio=[1.0, 1.0], previous_residual=None)
    try:
        with pytest.raises(WanHipError):
            model([lats[3].cuda()], t=t, context=[ctx.cuda()], real_step_no=1)          # would skip, nothing stored yet
        c.accumulated_err, c.accumulated_steps, c.accumulated_ratio = [0.0, 0.0], [0, 0], [1.0, 1.0]
        a = model([lats[3].cuda()], t=t, context=[ctx.cuda()], real_step_no=0)[0]    # computes, stores the residual
        assert torch.equal(a, plain)
        b = model([lats[3].cuda()], t=t, context=[ctx.cuda()], real_step_no=1)[0]    # skips
        # (the fp32 residual of the mixed plan gives x back up to one fp32 rounding: nearly always the same output)
        assert c.skipped_steps == 2 and rel(b, a.cpu()) < 1e-2 and (model.mixed_precision or not torch.equal(a, b))
        assert c.previous_residual[0].dtype == (torch.float32 if model.mixed_precision else torch.bfloat16)
    finally:
        model.cache = None


def test_generate_with_magcache(model):
    from wan2gp_amd.pipeline import WanAny2VHIP
    _, _, ctx, ctx_null = inputs(CFG)
    pipe = WanAny2VHIP(model)
    run = lambda: pipe.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), width=64, height=64, frame_num=5, sampling_steps=8,
                                guide_scale=3.0, seed=11, return_latents=True)["latents"].cpu()
    base = run()
    c = model.cache = mk("mag")
    try:
        fast = run()
        assert c.skipped_steps > 0 and torch.isfinite(fast).all()
        d = ((fast - base).norm() / base.norm()).item()
        print(f"generate with MagCache x{c.multiplier}: skipped {c.skipped_steps}/8, latents differ by {d:.3f} (relative)")
        assert 0 < d < 0.5
    finally:
        model.cache = None
