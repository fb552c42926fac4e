"""CPU: the control flow of `WanAny2VHIP.generate` (mirror of models/wan/any2video.py:1490-1750) with the two HIP entry points
it touches outside the model -- `ops.lincomb` (scheduler updates) and `ops.cfg_combine` -- replaced by torch one-liners IN THE
TEST and a fake DiT that records its calls.  No arithmetic claim is made here (the kernels are checked by the -m gpu suite);
what is checked is the host logic around them: which expert runs which step, which keyword arguments it receives, when the
step-skipping cache / LoRA multipliers / guidance variants / RIFLEx tables are engaged, for every solver."""
import types

import pytest
import torch

from wan2gp_amd import ops, pipeline, schedulers
from wan2gp_amd.pipeline import WanAny2VHIP


class FakeDiT:
    """Stands in for WanModelHIP: consumes the x list, returns one fp32 prediction per stream, records the call."""

    def __init__(self, tag, out_dim=16):
        self.tag, self.out_dim, self.calls, self.cache, self.loras = tag, out_dim, [], None, None
        self.device = torch.device("cpu")

    def __call__(self, x, t, context, **kw):
        xs = list(x)
        x.clear()
        tmax = float(t.flatten().max())                      # per-frame t (ti2v injection): the denoising frames' timestep
        self.calls.append(dict(tag=self.tag, t=tmax, n=len(xs), step=kw.get("current_step_no"), x_id=kw.get("x_id", 0),
                               freqs=kw["freqs"], y=kw.get("y")))
        g = torch.Generator().manual_seed(int(tmax) + len(self.calls))
        return [0.1 * torch.randn(u.shape, generator=g) + 0.05 * i for i, u in enumerate(xs)]

    # step-skipping cache API used by generate()
    def compute_magcache_threshold(self, start_step, timesteps=None, speed_factor=0):
        from wan2gp_amd import skipcache
        self.thresholds = ("mag", start_step, len(timesteps), speed_factor)
        return skipcache.compute_magcache_threshold(self.cache, start_step, timesteps, speed_factor)


@pytest.fixture(autouse=True)
def torch_stubs(monkeypatch):
    def lincomb(tensors, coefs, out=None):
        r = sum(float(c) * t_.float() for c, t_ in zip(coefs, tensors))
        return r if out is None else out.copy_(r)
    monkeypatch.setattr(ops, "lincomb", lincomb)
    monkeypatch.setattr(ops, "cfg_combine", lambda c, u, g, out=None: u + g * (c - u))
    yield


def run(pipe, **kw):
    ctx = torch.zeros(1, 512, 4096, dtype=torch.bfloat16)
    args = dict(context=ctx, context_null=ctx, width=64, height=64, frame_num=9, sampling_steps=6, guide_scale=4.0, seed=5,
                return_latents=True)
    args.update(kw)
    return pipe.generate(**args)


@pytest.mark.parametrize("solver", ["unipc", "euler", "dpm++", "causvid", "lcm"])
def test_every_solver_drives_the_model_once_per_step(solver):
    m = FakeDiT("A")
    steps = 4 if solver == "lcm" else 6
    out = run(WanAny2VHIP(m, device="cpu"), sample_solver=solver, sampling_steps=steps)
    assert torch.isfinite(out["latents"]).all() and tuple(out["latents"].shape) == (1, 16, 3, 8, 8)
    assert len(m.calls) >= min(steps, 3) and all(c["n"] == 2 for c in m.calls)                  # joint CFG pass
    assert [c["step"] for c in m.calls] == list(range(len(m.calls)))
    ts = [c["t"] for c in m.calls]
    assert ts == sorted(ts, reverse=True)


def test_two_experts_switch_at_the_threshold_and_guidance_scale_follows():
    a, b = FakeDiT("A"), FakeDiT("B")
    seen = []
    orig = ops.cfg_combine
    ops.cfg_combine = lambda c, u, g, out=None: (seen.append(g), orig(c, u, g))[1]
    try:
        run(WanAny2VHIP(a, b, device="cpu"), guide2_scale=2.5, guide_phases=2, switch_threshold=800, model_switch_phase=1)
    finally:
        ops.cfg_combine = orig
    assert a.calls and b.calls and all(c["t"] > 800 for c in a.calls) and all(c["t"] <= 800 for c in b.calls)
    assert seen == [4.0] * len(a.calls) + [2.5] * len(b.calls)
    assert len(a.calls) + len(b.calls) == 6


def test_single_passes_when_joint_pass_is_off_and_no_cfg_at_scale_one():
    m = FakeDiT("A")
    run(WanAny2VHIP(m, device="cpu"), joint_pass=False)
    assert all(c["n"] == 1 for c in m.calls) and [c["x_id"] for c in m.calls[:4]] == [0, 1, 0, 1] and len(m.calls) == 12
    m = FakeDiT("A")
    run(WanAny2VHIP(m, device="cpu"), guide_scale=1.0)
    assert len(m.calls) == 6 and all(c["n"] == 1 for c in m.calls)


def test_guidance_variants_and_riflex_are_engaged(monkeypatch):
    from wan2gp_amd import guidance
    used = []
    real = guidance.combine
    monkeypatch.setattr(guidance, "combine", lambda *a, **k: (used.append(a[3:7]), real(*a, **k))[1])
    m = FakeDiT("A")
    run(WanAny2VHIP(m, device="cpu"), cfg_star_switch=1, cfg_zero_step=1)
    assert [u[0] for u in used] == list(range(6)) and all(u[1] == 0 and u[2] == 1 and u[3] == 1 for u in used)
    used.clear()
    run(WanAny2VHIP(FakeDiT("A"), device="cpu"), apg_switch=1)
    assert len(used) == 6 and all(u[1] == 1 for u in used)
    plain = FakeDiT("A"); run(WanAny2VHIP(plain, device="cpu"))
    rif = FakeDiT("A"); run(WanAny2VHIP(rif, device="cpu"), enable_RIFLEx=True)
    assert not torch.equal(plain.calls[0]["freqs"][0], rif.calls[0]["freqs"][0])


def test_magcache_is_reset_and_thresholded_before_the_loop():
    from wan2gp_amd.skipcache import SkipStepsCache
    m = FakeDiT("A")
    m.cache = SkipStepsCache(cache_type="mag", multiplier=2.0, start_step=1, magcache_K=2, magcache_thresh=0,
                             def_mag_ratios=[0.99] * 10, skipped_steps=7, previous_residual="stale")
    run(WanAny2VHIP(m, device="cpu"))
    c = m.cache
    assert m.thresholds == ("mag", 1, 6, 2.0) and c.num_steps == 6 and c.previous_residual == [None, None]
    assert c.skipped_steps == 0 and len(c.mag_ratios) == 12 and c.accumulated_steps == [0, 0] and c.one_for_all is False


def test_lora_multipliers_follow_the_phase_of_each_expert():
    from wan2gp_amd.lora import parse_loras_multipliers

    class Rec:
        def __init__(self):
            self.steps = []

        def set_step(self, slists, n, step_no, s1, s2):
            from wan2gp_amd.lora import step_multipliers
            self.steps.append((step_no, step_multipliers(slists, n, step_no, s1, s2)))
    a, b = FakeDiT("A"), FakeDiT("B")
    a.loras, b.loras = Rec(), Rec()
    _, slists, err = parse_loras_multipliers("1;0 0;1", 2, 6, nb_phases=2)
    assert err == ""
    run(WanAny2VHIP(a, b, device="cpu"), guide_phases=2, switch_threshold=800, loras_slists=slists)
    assert a.loras.steps and all(m == [1.0, 0.0] for _, m in a.loras.steps)
    assert b.loras.steps and all(m == [0.0, 1.0] for _, m in b.loras.steps)
    assert [s for s, _ in a.loras.steps + b.loras.steps] == list(range(6))


def test_vace_control_video_reaches_the_model_as_context():
    from oracle.make_golden_vace_context import FakeVAE, inputs

    class VaceDiT(FakeDiT):
        def __call__(self, x, t, context, vace_context=None, vace_context_scale=None, **kw):
            self.vace = (None if vace_context is None else [tuple(z.shape) for z in vace_context], vace_context_scale)
            return super().__call__(x, t, context, **kw)
    frames, mask, _ = inputs()
    m = VaceDiT("A")
    run(WanAny2VHIP(m, vae=FakeVAE(), device="cpu"), width=48, height=32, input_frames=frames, input_masks=mask, context_scale=[0.8])
    assert m.vace == ([(96, 3, 4, 6)], [0.8]) and len(m.calls) == 6
    m = VaceDiT("A")
    run(WanAny2VHIP(m, vae=FakeVAE(), device="cpu"))
    assert m.vace == (None, None)


def test_sub_parallel_windows_run_every_step_on_overlapping_latent_windows():
    """frame_num 41 -> 11 latent frames; window 17 px-frames / overlap 5 -> 5 / 2 latent frames -> windows (0,5) (3,8) (6,11):
    three forwards per step, each on its window plus one anchor frame, with the RoPE rows of exactly those frames; the
    step-skipping cache is parked for the run and restored afterwards (any2video.py:1392-1397, :1448-1462)."""
    from wan2gp_amd.skipcache import SkipStepsCache
    m = FakeDiT("A")
    cache = m.cache = SkipStepsCache(cache_type="mag", multiplier=2.0, start_step=1, magcache_K=2, magcache_thresh=0, def_mag_ratios=[0.99] * 10,
                                     previous_residual="x", previous_modulated_input="y")
    seen = []
    orig = m.__class__.__call__

    def spy(self, x, t, context, **kw):
        seen.append((x[0].shape[2], kw["freqs"][0].shape[0], self.cache))
        return orig(self, x, t, context, **kw)
    m.__class__ = type("Spy", (FakeDiT,), {"__call__": spy})
    out = run(WanAny2VHIP(m, device="cpu"), frame_num=41, sampling_steps=3, sub_parallel_window_size=17, sub_parallel_window_overlap=5)
    assert tuple(out["latents"].shape) == (1, 16, 11, 8, 8) and torch.isfinite(out["latents"]).all()
    tok = 4 * 4
    assert [s[:2] for s in seen] == [(5, 5 * tok), (6, 6 * tok), (6, 6 * tok)] * 3           # anchor frame in front of windows 2 and 3
    assert all(s[2] is None for s in seen)                                                   # cache parked while windows run
    assert m.cache is cache and cache.previous_residual is None and cache.previous_modulated_input is None
    # a window that covers the clip is no window at all
    m2 = FakeDiT("A")
    run(WanAny2VHIP(m2, device="cpu"), frame_num=9, sampling_steps=2, sub_parallel_window_size=81)
    assert len(m2.calls) == 2


def test_three_guidance_phases_switch_expert_at_the_second_boundary():
    """guide_phases 3, model_switch_phase 2 (any2video.py:1437-1443, :1491-1492): expert 1 keeps running through phase 2 with
    guide2_scale, expert 2 takes over at switch2_threshold with guide3_scale."""
    a, b = FakeDiT("A"), FakeDiT("B")
    seen = []
    orig = ops.cfg_combine
    ops.cfg_combine = lambda c, u, g, out=None: (seen.append(g), orig(c, u, g))[1]
    try:
        run(WanAny2VHIP(a, b, device="cpu"), sampling_steps=8, guide2_scale=3.0, guide3_scale=2.0, guide_phases=3, switch_threshold=900,
            switch2_threshold=600, model_switch_phase=2)
    finally:
        ops.cfg_combine = orig
    ta, tb = [c["t"] for c in a.calls], [c["t"] for c in b.calls]
    assert ta and tb and min(ta) > 600 and max(tb) <= 600 and len(ta) + len(tb) == 8
    n1 = sum(1 for t in ta if t > 900)
    assert seen == [4.0] * n1 + [3.0] * (len(ta) - n1) + [2.0] * len(tb) and 0 < n1 < len(ta)


def test_i2v_start_image_conditions_every_step_and_is_restored_at_the_end():
    """image_start: y = [mask ; VAE latents] reaches every forward, the known first latent is re-noised before each step and
    written back clean after the last one (any2video.py:699-782, :1517-1523, :1755-1756)."""
    from oracle.make_golden_i2v_cond import FakeVAE
    m = FakeDiT("A")
    pipe = WanAny2VHIP(m, vae=FakeVAE(), device="cpu")
    img = torch.rand(3, 32, 48) * 2 - 1
    out = run(pipe, width=48, height=32, frame_num=9, sampling_steps=3, image_start=img)
    assert all(c["y"] is not None and tuple(c["y"].shape) == (20, 3, 4, 6) for c in m.calls) and len(m.calls) == 3
    y, ext = pipe.build_i2v_conditioning(img, 9, 32, 48)
    assert torch.equal(m.calls[0]["y"], y) and torch.equal(out["latents"][:, :, :1], ext)
    # prefix video of 5 frames: two known latent frames
    vid = torch.rand(3, 5, 32, 48) * 2 - 1
    out = run(WanAny2VHIP(FakeDiT("A"), vae=FakeVAE(), device="cpu"), width=48, height=32, frame_num=13, sampling_steps=2, image_start=vid,
              motion_amplitude=1.3)
    _, ext = pipe.build_i2v_conditioning(vid, 13, 32, 48, 0, 1.3)
    assert ext.shape[2] == 2 and torch.equal(out["latents"][:, :, :2], ext)


# ---- round-2 advisor items ---------------------------------------------------------------------------------------------------
def test_text_encoder_output_is_padded_like_the_reference():
    """any2video.py:587-593: the UMT5 wrapper returns unpadded [n_tokens, 4096]; generate() zero-pads to text_len rows and
    adds the batch axis before the DiT sees it."""
    seen = []

    class Dit(FakeDiT):
        text_len = 512

        def __call__(self, x, t, context, **kw):
            seen.append([tuple(c.shape) for c in context] + [c.dtype for c in context] + [float(c[0, 100:].abs().sum()) for c in context])
            return super().__call__(x, t, context, **kw)
    enc_calls = []

    def encoder(prompts, device):
        enc_calls.append(list(prompts))
        n = 7 if prompts[0] else 1
        return [torch.ones(n, 4096)]                          # fp32, unpadded, like T5EncoderModel.__call__ (t5.py:709-716)
    pipe = WanAny2VHIP(Dit("A"), device="cpu", text_encoder=encoder)
    out = pipe.generate(input_prompt="a cat", n_prompt="", width=64, height=64, frame_num=9, sampling_steps=2, guide_scale=4.0,
                        seed=1, return_latents=True)
    assert out is not None and enc_calls == [["a cat"], [""]]
    assert seen[0] == [(1, 512, 4096), (1, 512, 4096), torch.bfloat16, torch.bfloat16, 0.0, 0.0]
    # the next window of the same video: both prompts come from the cache (TextEncoderCache, any2video.py:589-592); a new prompt is encoded
    first = seen[0]
    pipe.generate(input_prompt="a cat", n_prompt="", width=64, height=64, frame_num=9, sampling_steps=1, guide_scale=4.0, seed=1, return_latents=True)
    assert enc_calls == [["a cat"], [""]] and seen[-1] == first
    pipe.generate(input_prompt="a dog", n_prompt="", width=64, height=64, frame_num=9, sampling_steps=1, guide_scale=4.0, seed=1, return_latents=True)
    assert enc_calls == [["a cat"], [""], ["a dog"]]


def test_shared_cache_object_is_configured_once_from_model():
    """wgp.py hands ONE cache object to both experts; the reference configures it once, from self.model (any2video.py:1396-1406)."""
    from wan2gp_amd.skipcache import SkipStepsCache
    a, b = FakeDiT("A"), FakeDiT("B")
    a.cache = b.cache = SkipStepsCache(cache_type="mag", multiplier=2.0, start_step=1, magcache_K=2, magcache_thresh=0,
                                       def_mag_ratios=[0.99] * 10, previous_residual=None)
    run(WanAny2VHIP(a, b, device="cpu"), guide_phases=2, switch_threshold=800, model_switch_phase=1)
    assert hasattr(a, "thresholds") and not hasattr(b, "thresholds")


def test_parked_caches_come_back_when_a_forward_raises():
    class Boom(FakeDiT):
        def __call__(self, x, t, context, **kw):
            raise RuntimeError("kernel failure")
    m = Boom("A")
    m.cache = types.SimpleNamespace(cache_type="mag", previous_residual=[1], previous_modulated_input=2)
    with pytest.raises(RuntimeError):
        run(WanAny2VHIP(m, device="cpu"), frame_num=17, sub_parallel_window_size=9, sub_parallel_window_overlap=1)
    assert m.cache is not None and m.cache.previous_residual is None


def test_clip_fea_reaches_the_model_and_i2v21_requires_it():
    got = []

    class Dit(FakeDiT):
        model_type = "i2v"

        def __call__(self, x, t, context, **kw):
            got.append(kw.get("clip_fea"))
            return super().__call__(x, t, context, **kw)
    pipe = WanAny2VHIP(Dit("A"), device="cpu")
    with pytest.raises(ValueError, match="clip_fea"):
        run(pipe, y=torch.zeros(20, 3, 8, 8))
    cf = torch.zeros(1, 257, 1280)
    run(pipe, y=torch.zeros(20, 3, 8, 8), clip_fea=cf, sampling_steps=2)
    assert got and all(g is cf for g in got)


def test_ti2v_timestep_injection_passes_a_per_frame_t_and_pins_the_source_latents():
    """any2video.py:1496-1499, :1753-1754: with an input video the ti2v model sees t = [0 (source frames), t, t, ...] and the
    source latents are re-imposed before every step and after the last."""
    seen = []

    class Dit(FakeDiT):
        model_type = "ti2v2_2"

        def __call__(self, x, t, context, **kw):
            seen.append((t.clone(), x[0][:, :, :1].clone()))
            return super().__call__(x, t, context, **kw)

    class Vae:
        def encode(self, videos, tile_size=0):
            return [torch.full((16, 1, 8, 8), 7.0)]
    out = run(WanAny2VHIP(Dit("A"), vae=Vae(), device="cpu"), input_video=torch.zeros(3, 1, 64, 64), sampling_steps=3)
    for t, first in seen:
        assert t.shape == (3,) and t[0] == 0 and (t[1:] == t[1]).all() and t[1] > 0
        assert (first == 7.0).all()
    assert (out["latents"][:, :, :1] == 7.0).all()
    # with sub-parallel windows the per-frame t follows each window (any2video.py:1303-1304): windows (0,2) and anchor + (1,3)
    seen.clear()
    run(WanAny2VHIP(Dit("A"), vae=Vae(), device="cpu"), input_video=torch.zeros(3, 1, 64, 64), sampling_steps=1, sub_parallel_window_size=5,
        sub_parallel_window_overlap=1)
    assert [tuple(t.shape) for t, _ in seen] == [(2,), (3,)] and all(t[0] == 0 and t[1] > 0 for t, _ in seen)
    # a model that is neither the 5B nor i2v: the reference reads the size of input_video and nothing else (any2video.py:571)
    seen.clear()
    run(WanAny2VHIP(FakeDiT("A"), vae=Vae(), device="cpu"), input_video=torch.zeros(3, 1, 64, 64))
    assert not seen


def test_nag_stacks_the_negative_prompt_under_the_positive_one_and_arms_both_experts():
    """any2video.py:607-608: NAG_scale > 1 -> context = cat([context, context_null]) and the three parameters reach the model
    (offload.shared_state there, model.nag here); with CFG on top the uncond stream keeps its plain batch-1 context."""
    class NagDiT(FakeDiT):
        nag = None

        def __call__(self, x, t, context, **kw):
            self.ctx_shapes = [tuple(c.shape) for c in context]
            self.ctx_rows = [c[:, 0, 0].float().tolist() for c in context]
            return super().__call__(x, t, context, **kw)
    a, b = NagDiT("A"), NagDiT("B")
    pos = torch.full((1, 512, 4096), 1.0, dtype=torch.bfloat16)
    neg = torch.full((1, 512, 4096), -1.0, dtype=torch.bfloat16)
    run(WanAny2VHIP(a, b, device="cpu"), context=pos, context_null=neg, guide_scale=1.0, NAG_scale=11, NAG_tau=2.5, NAG_alpha=0.25,
        guide_phases=2, guide2_scale=1.0, switch_threshold=800)
    assert a.nag == b.nag == (11.0, 2.5, 0.25)
    assert a.ctx_shapes == b.ctx_shapes == [(2, 512, 4096)] and a.ctx_rows == [[1.0, -1.0]]
    m = NagDiT("A")
    run(WanAny2VHIP(m, device="cpu"), context=pos, context_null=neg, NAG_scale=11)
    assert m.ctx_shapes == [(2, 512, 4096), (1, 512, 4096)] and m.ctx_rows == [[1.0, -1.0], [-1.0]] and m.nag == (11.0, 3.5, 0.5)
    m = NagDiT("A")
    run(WanAny2VHIP(m, device="cpu"), context=pos, context_null=neg, NAG_scale=1)        # off (nag_scale <= 1, model.py:260)
    assert m.ctx_shapes == [(1, 512, 4096), (1, 512, 4096)] and m.nag is None


def test_image_end_adds_a_latent_frame_only_for_the_wan21_i2v_model_and_trims_it():
    """any2video.py:684-691, :1759: start + end image.  Wan2.2 i2v (i2v2_2): same latent frame count, the mask's last entry is 1;
    Wan2.1 i2v: one extra latent frame through the whole loop, encoded with any_end_frame, cut off before decoding."""
    from oracle.make_golden_i2v_cond import FakeVAE

    class Vae(FakeVAE):
        def __init__(self):
            self.encodes = []

        def encode(self, videos, tile_size=0, any_end_frame=False):
            self.encodes.append((tuple(videos[0].shape), any_end_frame))
            return super().encode(videos, tile_size, any_end_frame)

    img, end = torch.rand(3, 64, 64) * 2 - 1, torch.rand(3, 64, 64) * 2 - 1
    for mt, lat_f, flag in (("i2v2_2", 3, False), ("i2v", 4, True)):
        m, vae = FakeDiT("A"), Vae()
        m.model_type = mt
        out = run(WanAny2VHIP(m, vae=vae, device="cpu"), image_start=img, image_end=end, clip_fea=torch.zeros(1, 257, 1280))
        assert vae.encodes == [((3, 10 if flag else 9, 64, 64), flag)]
        ys = [c["y"] for c in m.calls]
        assert all(tuple(y.shape) == (20, lat_f, 8, 8) for y in ys)
        # the mask folds 4 frames into 4 channels per latent frame: Wan2.2 marks the clip's last FRAME (channel 3 of the last latent
        # frame), the Wan2.1 form repeats the end frame's entry over all four channels of its own latent frame (:749)
        last = ys[0][:4, -1]
        assert bool((last[3] == 1).all()) and bool((last[:3] == (1 if flag else 0)).all()) and bool((ys[0][:4, 1:-1] == 0).all())
        assert bool((ys[0][:4, 0] == 1).all())
        assert tuple(out["latents"].shape) == (1, 16, 3, 8, 8)                                 # the extra frame is trimmed (:1759)
    with pytest.raises(ValueError):
        run(WanAny2VHIP(FakeDiT("A"), device="cpu"), image_end=end)


def test_return_latent_slice_hands_back_the_requested_latent_frames():
    """any2video.py:1760-1761, :1810: `return_latent_slice` (a slice over the latent time axis, what a sliding-window caller
    overlaps the next window with) is cut after the end-frame trim and returned beside the video."""
    out = run(WanAny2VHIP(FakeDiT("A"), device="cpu"), return_latent_slice=slice(-2, None))
    assert tuple(out["latent_slice"].shape) == (1, 16, 2, 8, 8) and torch.equal(out["latent_slice"], out["latents"][:, :, -2:])
    assert run(WanAny2VHIP(FakeDiT("A"), device="cpu"))["latent_slice"] is None


def test_video_to_video_cuts_the_schedule_or_reinjects_and_pins_unmasked_regions():
    """"G" in video_prompt_type (any2video.py:1004-1044, :1504-1515, :1737-1740; the arithmetic itself is pinned to the reference's
    statements in tests/test_v2v_vs_golden.py): what generate() does around it -- the Python scheduler mirror with its tables cut
    short, real_step_no offset, the first forward seeing the noised source, the re-injection path for a source shorter than the
    clip, the masked merge behind the scheduler step."""
    from oracle.make_golden_i2v_cond import FakeVAE

    class Rec(FakeDiT):
        def __call__(self, x, t, context, **kw):
            self.seen = getattr(self, "seen", []) + [(x[0].clone(), kw.get("real_step_no"))]
            return super().__call__(x, t, context, **kw)

    g = torch.Generator().manual_seed(0)
    vid = torch.rand(3, 9, 64, 64, generator=g) * 2 - 1
    vae = FakeVAE()
    src = vae.encode([vid])[0].unsqueeze(0)
    # (a) the source covers the clip: 6 steps at strength 0.5 -> the last 3 steps only, starting from the noised source
    m = Rec("A")
    out = run(WanAny2VHIP(m, vae=vae, device="cpu"), input_frames=vid, video_prompt_type="G", denoising_strength=0.5, guide_scale=1.0)
    assert len(m.calls) == 3 and [r for _, r in m.seen] == [3, 4, 5]
    ts_all = schedulers.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    ts_all.set_timesteps(6, device="cpu", shift=5.0)
    assert [c["t"] for c in m.calls] == [float(v) for v in ts_all.timesteps[3:]]
    x0, sigma = m.seen[0][0], float(ts_all.timesteps[3]) / 1000
    noise = (x0 - (1 - sigma) * src) / sigma                              # what must have been the initial noise
    assert abs(float(noise.mean())) < 0.05 and abs(float(noise.std()) - 1.0) < 0.05
    assert torch.isfinite(out["latents"]).all()
    # (b) without "G" the same arguments are a plain generation (and input_frames without masks is refused as a VACE input)
    with pytest.raises(ValueError):
        run(WanAny2VHIP(FakeDiT("A"), vae=vae, device="cpu"), input_frames=vid, denoising_strength=0.5)
    # (c) a source shorter than the clip: every step runs, the first injection steps see the noised source in front
    m = Rec("A")
    run(WanAny2VHIP(m, vae=vae, device="cpu"), input_frames=vid[:, :5], video_prompt_type="G", denoising_strength=0.5, guide_scale=1.0)
    assert len(m.calls) == 6 and [r for _, r in m.seen] == list(range(6))
    src5 = vae.encode([vid[:, :5]])[0].unsqueeze(0)
    for k in range(4):                                                    # steps 0..3 (i <= injection step 3) are re-injected
        sg = float(ts_all.timesteps[k]) / 1000
        z = (m.seen[k][0][:, :, :2] - (1 - sg) * src5) / sg
        assert torch.allclose(z, (m.seen[0][0][:, :, :2] - (1 - float(ts_all.timesteps[0]) / 1000) * src5) / (float(ts_all.timesteps[0]) / 1000),
                              atol=1e-4)                                  # always the same initial noise
    # (d) a mask: outside it the result of the masked steps is the source at the next step's noise level
    mask = torch.zeros(1, 9, 64, 64)
    mask[..., :32] = 1                                                    # regenerate the left half only
    m = Rec("A")
    out = run(WanAny2VHIP(m, vae=vae, device="cpu"), input_frames=vid, input_masks=mask, video_prompt_type="G", denoising_strength=0.5,
              masking_strength=1.0, guide_scale=1.0)
    right = out["latents"][..., 4:]
    assert torch.allclose(right, src[..., 4:], atol=1e-5)                 # last step: sigma_next = 0 -> exactly the source
    assert not torch.allclose(out["latents"][..., :4], src[..., :4], atol=1e-2)


def test_nag_parameters_reach_a_model_driven_by_the_references_own_generate(monkeypatch):
    """The reference publishes NAG through mmgp.offload.shared_state["_nag_*"] (any2video.py:607); a WanModelHIP used under the
    reference's own generate() has no `.nag` set and reads them from there (WanModelHIP._nag_params)."""
    import sys
    import types as _t
    from wan2gp_amd.model import WanModelHIP
    m = object.__new__(WanModelHIP)                                     # no library: only the parameter lookup is exercised
    m.nag = None
    assert m._nag_params() is None
    off = _t.ModuleType("mmgp.offload")
    off.shared_state = {"_nag_scale": 11, "_nag_tau": 3.5, "_nag_alpha": 0.5}
    monkeypatch.setitem(sys.modules, "mmgp.offload", off)
    assert m._nag_params() == (11.0, 3.5, 0.5)
    off.shared_state["_nag_scale"] = 1                                  # nag_scale <= 1: off (model.py:260)
    assert m._nag_params() is None
    m.nag = (2.0, 2.5, 0.25)                                            # an explicit setting wins
    assert m._nag_params() == (2.0, 2.5, 0.25)


def test_self_refiner_repeats_the_planned_steps_through_the_same_denoise_function():
    """any2video.py:1485-1488, :1729-1731 (the handler itself is pinned to the reference's in tests/test_refiner_vs_reference.py):
    generate() builds it from the four self_refiner_* keywords, switches to a copyable scheduler, and the planned steps call the
    model again through the same CFG path (joint pass, both contexts)."""
    m = FakeDiT("A")
    out = run(WanAny2VHIP(m, device="cpu"), self_refiner_setting=1, self_refiner_plan="1-2:3", self_refiner_f_uncertainty=0.0)
    assert torch.isfinite(out["latents"]).all()
    assert len(m.calls) == 6 + 2 * 2 and all(c["n"] == 2 for c in m.calls)             # two extra repetitions on steps 1 and 2
    assert [c["step"] for c in m.calls] == [0, 1, 1, 1, 2, 2, 2, 3, 4, 5]
    m = FakeDiT("A")
    run(WanAny2VHIP(m, device="cpu"), self_refiner_setting=0, self_refiner_plan="1-2:3")
    assert len(m.calls) == 6


def test_skip_layer_guidance_is_passed_inside_its_window_of_steps():
    """any2video.py:1502: kwargs["perturbation_layers"] = perturbation_layers if int(start * steps) <= i < int(end * steps) else None."""
    class Rec(FakeDiT):
        def __call__(self, x, t, context, **kw):
            self.slg = getattr(self, "slg", []) + [kw.get("perturbation_layers")]
            return super().__call__(x, t, context, **kw)
    m = Rec("A")
    run(WanAny2VHIP(m, device="cpu"), perturbation_layers=[9, 10], perturbation_start=0.2, perturbation_end=0.7)
    assert m.slg == [None, [9, 10], [9, 10], [9, 10], None, None]
    m = Rec("A")
    run(WanAny2VHIP(m, device="cpu"))
    assert m.slg == [None] * 6


def test_vace_sliding_window_pins_the_overlap_of_the_control_video():
    """Sliding windows on the VACE path (any2video.py:1150-1152, :1517-1526, :1755-1756): with the previous window's
    `overlapped_latents` [1,16,n,h,w] the INACTIVE half of the context's first n latent frames is the pinned prefix -- in front of every
    step latents[:, :, :n] = prefix (1 - t/1000) + noise t/1000 and context[0:16, :n] = prefix (1 - overlap_noise/1000) + noise
    overlap_noise/1000 (two draws from the global generator, in this order), behind the last step the clean prefix; the requested
    latent slice goes back to the caller for the next window.  Without overlapped latents nothing is pinned; on a model without
    VACE blocks (the i2v path of the reference has the use switched off, :779) the keyword changes nothing."""
    from oracle.make_golden_vace_context import FakeVAE, inputs

    class VaceDiT(FakeDiT):
        def __call__(self, x, t, context, vace_context=None, **kw):
            self.seen = getattr(self, "seen", [])
            self.seen.append((float(t[0] if torch.is_tensor(t) and t.dim() else t), x[0][:, :, :2].clone(), vace_context[0][:16, :2].clone()))
            return super().__call__(x, t, context, **kw)
    frames, mask, _ = inputs()
    m = VaceDiT("A")
    pipe = WanAny2VHIP(m, vae=FakeVAE(), device="cpu")
    prefix = pipe.vace_context([frames], [mask], None, 0)[0][:16, :2].clone()
    torch.manual_seed(77)
    out = run(pipe, width=48, height=32, input_frames=frames, input_masks=mask, overlapped_latents=torch.zeros(1, 16, 2, 4, 6), overlap_noise=20,
              return_latent_slice=slice(-2, None), joint_pass=True)
    assert torch.equal(out["latents"][:, :, :2], prefix.unsqueeze(0))                     # clean behind the last step
    assert torch.equal(out["latent_slice"], out["latents"][:, :, -2:])
    torch.manual_seed(77)
    steps = []
    for t, xs, zz in m.seen:                                                              # joint pass: one call per step
        if not steps or steps[-1][0] != t:
            steps.append((t, xs, zz))
    assert len(steps) == 6
    ext = prefix.unsqueeze(0)
    for t, xs, zz in steps:
        f = t / 1000.0
        want_x = ext * (1.0 - f) + torch.randn_like(ext) * f
        want_z = ext[0] * (1.0 - 0.02) + torch.randn_like(ext[0]) * 0.02
        assert torch.equal(xs, want_x) and torch.equal(zz, want_z), t
    # no overlapped latents: the context's first frames are left alone and nothing is pinned
    m2 = VaceDiT("A")
    out2 = run(WanAny2VHIP(m2, vae=FakeVAE(), device="cpu"), width=48, height=32, input_frames=frames, input_masks=mask)
    assert all(torch.equal(zz, prefix) for _, _, zz in m2.seen) and not torch.equal(out2["latents"][:, :, :2], ext)
    # a model without VACE blocks: accepted and without effect (wgp.py passes it for every window after the first)
    a = run(WanAny2VHIP(FakeDiT("A"), device="cpu"), overlapped_latents=torch.zeros(1, 16, 2, 8, 8), overlap_noise=20)
    b = run(WanAny2VHIP(FakeDiT("A"), device="cpu"))
    assert torch.equal(a["latents"], b["latents"])


def test_vace_second_control_video_is_a_second_context_with_its_own_scale():
    """any2video.py:1129-1130, :1148: input_frames2 / input_masks2 = one more VACE context through the same context blocks;
    context_scale carries one weight per context (wgp.py:7525).  Outside the VACE path the keyword is refused."""
    from oracle.make_golden_vace_context import FakeVAE, inputs

    class VaceDiT(FakeDiT):
        vace_layers = (0,)

        def __call__(self, x, t, context, vace_context=None, vace_context_scale=None, **kw):
            self.vace = ([tuple(z.shape) for z in vace_context], list(vace_context_scale), [z.clone() for z in vace_context])
            return super().__call__(x, t, context, **kw)
    frames, mask, _ = inputs()
    frames2, mask2 = frames.flip(1), 1 - mask
    m = VaceDiT("A")
    pipe = WanAny2VHIP(m, vae=FakeVAE(), device="cpu")
    run(pipe, width=48, height=32, input_frames=frames, input_masks=mask, input_frames2=frames2, input_masks2=mask2, context_scale=[0.5, 0.25],
        control_scale_alt=0.3)
    assert m.vace[0] == [(96, 3, 4, 6), (96, 3, 4, 6)] and m.vace[1] == [0.5, 0.25]
    want = pipe.vace_context([frames, frames2], [mask, mask2], None, 0)
    assert torch.equal(m.vace[2][0], want[0]) and torch.equal(m.vace[2][1], want[1]) and not torch.equal(want[0], want[1])
    run(pipe, width=48, height=32, input_frames=frames, input_masks=mask, input_frames2=frames2, input_masks2=mask2)
    assert m.vace[1] == [1.0, 1.0]                                                                  # :1148
    with pytest.raises(NotImplementedError, match="input_frames2"):
        run(WanAny2VHIP(FakeDiT("A"), device="cpu"), input_frames2=frames2, input_masks2=mask2)
    with pytest.raises(ValueError, match="come together"):
        run(pipe, width=48, height=32, input_frames=frames, input_masks=mask, input_frames2=frames2)


def test_vace_reference_images_are_extra_latent_frames_in_front_and_are_cut_off_at_the_end():
    """VACE reference images (any2video.py:1128-1166, :1745, :1758, ref_images_before): n images = n extra latent frames in front of the
    control context (their latents beside zero masks, vace_encode_frames / _masks -- pinned to the reference by
    tests/golden/vace_context.npz) AND of the latents the model sees; the previews and the result leave them out.  Together with
    sliding-window overlap the pinned prefix spans reference frames + overlap, the context noise only the overlap (:1151-1152, :1526).
    Outside the VACE path: refused."""
    from oracle.make_golden_vace_context import FakeVAE, inputs

    class VaceDiT(FakeDiT):
        vace_layers = (0,)

        def __call__(self, x, t, context, vace_context=None, **kw):
            self.seen = getattr(self, "seen", [])
            self.seen.append((tuple(x[0].shape), tuple(vace_context[0].shape), vace_context[0][:16, :4].clone()))
            return super().__call__(x, t, context, **kw)
    frames, mask, refs = inputs()
    m = VaceDiT("A")
    pipe = WanAny2VHIP(m, vae=FakeVAE(), device="cpu")
    previews = []
    out = run(pipe, width=48, height=32, input_frames=frames, input_masks=mask, input_ref_images=refs, input_ref_masks=[None, None],
              callback=lambda i, lat=None, *a, **k: previews.append(None if lat is None else tuple(lat.shape)))
    assert all(xs == (1, 16, 3 + 2, 4, 6) and zs == (96, 3 + 2, 4, 6) for xs, zs, _ in m.seen)       # 3 latent frames + 2 reference frames
    assert tuple(out["latents"].shape) == (1, 16, 3, 4, 6)
    assert [p for p in previews if p is not None and len(p) == 4][-1] == (16, 3, 4, 6)
    zref = pipe.vace_context([frames], [mask], refs, 0)[0]
    assert torch.equal(m.seen[0][2][:, :2], zref[:16, :2]) and torch.equal(zref[16:32, :2], torch.zeros_like(zref[16:32, :2]))   # ref latents | zeros
    assert torch.equal(zref[32:, :2], torch.zeros_like(zref[32:, :2]))                               # zero mask frames in front
    # + sliding-window overlap of 1 latent frame: prefix = 2 reference frames + 1 overlap frame, context noise on the overlap frame only
    m2 = VaceDiT("A")
    torch.manual_seed(5)
    out2 = run(WanAny2VHIP(m2, vae=FakeVAE(), device="cpu"), width=48, height=32, input_frames=frames, input_masks=mask, input_ref_images=refs,
               overlapped_latents=torch.zeros(1, 16, 1, 4, 6), overlap_noise=20, return_latent_slice=slice(-1, None))
    assert tuple(out2["latents"].shape) == (1, 16, 3, 4, 6) and tuple(out2["latent_slice"].shape) == (1, 16, 1, 4, 6)
    assert torch.equal(out2["latents"][:, :, :1], zref[:16, 2:3].unsqueeze(0))                       # the clean overlap frame leads the result
    assert all(torch.equal(z4[:, :2], zref[:16, :2]) for _, _, z4 in m2.seen)                       # reference frames of the context untouched
    assert not any(torch.equal(z4[:, 2:3], zref[:16, 2:3]) for _, _, z4 in m2.seen)                  # its overlap frame re-noised every step
    # refusals
    with pytest.raises(NotImplementedError, match="VACE path"):
        run(WanAny2VHIP(FakeDiT("A"), device="cpu"), input_ref_images=refs)
    # a background mask for the first reference image (:1138-1145): its context frame becomes the masked encoding (pinned to the
    # reference's functions by tests/test_vace_context_vs_golden.py), nothing else changes
    bgm = (torch.rand(1, 1, 32, 48, generator=torch.Generator().manual_seed(78)) > 0.5).float()
    m3 = VaceDiT("A")
    run(WanAny2VHIP(m3, vae=FakeVAE(), device="cpu"), width=48, height=32, input_frames=frames, input_masks=mask, input_ref_images=refs,
        input_ref_masks=[bgm, None])
    zbg = pipe.vace_context([frames], [mask], refs, 0, [bgm, None])[0]
    assert all(xs == (1, 16, 5, 4, 6) and zs == (96, 5, 4, 6) and torch.equal(z4, zbg[:16, :4]) for xs, zs, z4 in m3.seen)
    assert not torch.equal(zbg[:16, :1], zref[:16, :1]) and torch.equal(zbg[:, 1:], zref[:, 1:])
    # together with sub-parallel windows (any2video.py:1222, :1234-1249, :1341-1343): the reference frames lead EVERY window
    # (latents, RoPE rows, context), arithmetic pinned to the reference's closures by tests/test_subparallel_vs_golden.py
    m4 = VaceDiT("A")
    out4 = run(WanAny2VHIP(m4, vae=FakeVAE(), device="cpu"), width=48, height=32, input_frames=frames, input_masks=mask, input_ref_images=refs,
               sub_parallel_window_size=5, sub_parallel_window_overlap=1, sampling_steps=2)
    assert tuple(out4["latents"].shape) == (1, 16, 3, 4, 6)
    shapes = [(xs[2], zs[1]) for xs, zs, _ in m4.seen[:2]]        # windows (0,2) and (1,3) of 3 latent frames: 2 + 2 frames, 2 + (anchor + 2)
    assert shapes == [(4, 4), (5, 5)], shapes
    assert all(torch.equal(z4[:, :2], zref[:16, :2]) for _, _, z4 in m4.seen)                        # the prefix of every window


def test_keywords_of_unserved_reference_paths_are_refused_not_ignored():
    """wgp.py passes every generate() the union of all variants' keywords (wgp.py:7762-7885): defaults and UI plumbing are accepted
    silently, a keyword that would change the video through a path this backend does not serve raises."""
    pipe = WanAny2VHIP(FakeDiT("A"), device="cpu")
    out = run(pipe, input_ref_images=None, audio_guide=None, overlap_noise=0, image_mode=0, alt_guide_scale=1.0, fit_into_canvas=True, window_no=1,
              offloadobj=object(), set_header_text=lambda *a: None, model_filename="x.safetensors", fps=16, gen_state={}, custom_settings=None)
    assert torch.isfinite(out["latents"]).all()
    for kw in (dict(input_faces=torch.zeros(3, 5, 8, 8)),
               dict(audio_proj=torch.zeros(1)), dict(image_mode=1), dict(alt_guide_scale=2.0), dict(vae_upsampler="x")):  # noqa: E501
        with pytest.raises(NotImplementedError, match=list(kw)[0]):
            run(pipe, **kw)


def _wgp_keywords(**over):
    """The keyword set of wgp.py:7762-7885 for a first-window Wan generation, with the values wgp.py gives them when the feature
    behind them is off: the hard-coded ones (`causal_block_size=5`, `causal_attention=True`), the UI defaults
    (`overlap_noise` = sliding_window_overlap_noise 20, `overlap_size` = reuse_frames), empty containers and None otherwise."""
    kw = dict(alt_prompt=None, image_start=None, image_end=None, input_frames=None, input_frames2=None, input_ref_images=None,
              input_ref_masks=None, input_masks=None, input_masks2=None, input_video=None, input_faces=None, input_custom=None,
              video_guide=None, video_guide2=None, denoising_strength=1.0, masking_strength=1.0, prefix_frames_count=0,
              batch_size=1, fit_into_canvas=1, shift=5.0, sample_solver="unipc", guide2_scale=3.0, guide3_scale=3.0,
              switch_threshold=0, switch2_threshold=0, guide_phases=1, model_switch_phase=1, embedded_guidance_scale=6.0,
              n_prompt="", callback=None, enable_RIFLEx=False, VAE_tile_size=0, joint_pass=True, perturbation_switch=0,
              perturbation_layers=[9], perturbation_start=0.1, perturbation_end=0.9, apg_switch=0, cfg_star_switch=0,
              cfg_zero_step=-1, alt_guide_scale=1.0, audio_cfg_scale=4.0, input_waveform=None, input_waveform_sample_rate=None,
              audio_guide=None, audio_guide2=None, audio_prompt_type="", audio_proj=None, audio_scale=None, audio_context_lens=None,
              context_scale=None, control_scale_alt=1.0, alt_scale=0.0, motion_amplitude=1.0, model_mode=None, causal_block_size=5,
              causal_attention=True, fps=16, overlapped_latents=None, return_latent_slice=None, overlap_noise=20, overlap_size=0,
              sub_parallel_window_size=0, sub_parallel_window_overlap=0, color_correction_strength=1.0, conditioning_latents_size=0,
              input_video_is_hdr=False, lora_dir="loras/wan", keep_frames_parsed=[], model_filename=["a.safetensors"],
              model_type="t2v", loras_slists=None, NAG_scale=1, NAG_tau=3.5, NAG_alpha=0.5, attention_sparsity=0,
              speakers_bboxes=None, image_mode=0, video_prompt_type="", window_no=1, offloadobj=object(),
              set_header_text=lambda *a: None, pre_video_frame=None, prefix_video=None, original_input_ref_images=[],
              image_refs_relative_size=50, outpainting_dims=None, face_arc_embeds=None, custom_settings=None,
              frame_window_options=None, gen_state={}, temperature=1.0, window_start_frame_no=0, input_video_strength=1.0,
              self_refiner_setting=0, self_refiner_plan="", self_refiner_f_uncertainty=0.0, self_refiner_certain_percentage=0.999,
              duration_seconds=5, pause_seconds=0, top_p=0.9, top_k=50, set_progress_status=lambda *a: None, loras_selected=[],
              frames_relative_positions_list=[], frames_to_inject=[], verbose_level=0, gen_cache=None, vae_upsampler=None,
              save_masks=False)
    kw.update(over)
    return kw


def test_generate_accepts_the_exact_keyword_set_wgp_passes_for_t2v():
    """Every wgp.py-driven call carries `causal_attention=True`, `overlap_noise=20`, ... (wgp.py:7826, :7830): a plain t2v run with
    that exact keyword set must run, and give the video the same call without the plumbing gives."""
    a, b = FakeDiT("A"), FakeDiT("A")
    got = run(WanAny2VHIP(a, device="cpu"), **_wgp_keywords(perturbation_layers=None))
    want = run(WanAny2VHIP(b, device="cpu"), shift=5.0)
    assert torch.equal(got["latents"], want["latents"]) and len(a.calls) == len(b.calls) == 6


class _StubVAE:
    """encode(): 16 latent channels, time (T - 1) // 4 + 1, space / 8, values a deterministic function of the frames."""

    def __init__(self):
        self.seen = []

    def encode(self, videos, tile_size=0, any_end_frame=False):
        self.seen.append(tuple(videos[0].shape))
        out = []
        for v in videos:
            T, H, W = v.shape[1:]
            t = (T - 1) // 4 + 1
            base = torch.nn.functional.adaptive_avg_pool3d(v[None].float(), (t, H // 8, W // 8))[0].mean(0, keepdim=True)
            out.append(base.repeat(16, 1, 1, 1) + torch.arange(16).view(16, 1, 1, 1) * 0.01)
        return out


def test_generate_takes_the_i2v_conditioning_from_input_video_like_the_reference():
    """any2video.py:671-680: the i2v path conditions on `input_video` (wgp.py passes input_video = pre_video_guide: the start image
    as [3, 1, H, W] or the video to continue, together with prefix_video / pre_video_frame / conditioning_latents_size > 0,
    wgp.py:7378-7394, :7714).  Same y and same result as image_start= (the direct-call spelling).  On a t2v-class
    model the reference reads nothing of input_video but its height and width (any2video.py:571): same here."""
    g = torch.Generator().manual_seed(3)
    img = torch.rand(3, 1, 64, 64, generator=g) * 2 - 1
    prefix = torch.rand(3, 5, 64, 64, generator=g) * 2 - 1
    for src in (img, prefix):
        a, b = FakeDiT("A"), FakeDiT("A")
        a.model_type = b.model_type = "i2v2_2"
        va, vb = _StubVAE(), _StubVAE()
        got = run(WanAny2VHIP(a, vae=va, device="cpu"),
                  **_wgp_keywords(input_video=src, prefix_video=src, pre_video_frame=src[:, -1], conditioning_latents_size=1,
                                  perturbation_layers=None, model_type="i2v_2_2"))
        want = run(WanAny2VHIP(b, vae=vb, device="cpu"), image_start=src, shift=5.0)
        assert va.seen == vb.seen == [(3, 9, 64, 64)]
        ya, yb = a.calls[0]["y"], b.calls[0]["y"]
        assert ya is not None and tuple(ya.shape) == (20, 3, 8, 8) and torch.equal(ya, yb)
        assert torch.equal(got["latents"], want["latents"])
    a = run(WanAny2VHIP(FakeDiT("A"), device="cpu"), input_video=img)                      # 64 x 64 like the default keywords: size unchanged
    b = run(WanAny2VHIP(FakeDiT("A"), device="cpu"))
    assert torch.equal(a["latents"], b["latents"])


def test_vace_second_sliding_window_gets_input_video_from_wgp_and_only_its_size_is_read():
    """Window 2 and later of a long VACE video as wgp.py drives them: input_video = pre_video_guide (the overlap frames of the previous
    window, wgp.py:7740, :7995) goes to EVERY model type together with overlapped_latents and prefix_frames_count.  The reference reads
    height / width from it (any2video.py:571) and nothing else on the VACE path -- the overlap itself arrives through input_frames and
    overlapped_latents (:837, :1150-1163).  Round 3 raised here ("input_video ... is the ti2v_2_2 conditioning path"), i.e. every long
    VACE video died after its first window.  Same result with and without the keyword."""
    from oracle.make_golden_vace_context import FakeVAE, inputs

    class VaceDiT(FakeDiT):
        vace_layers = (0,)

        def __call__(self, x, t, context, vace_context=None, **kw):
            assert vace_context is not None
            return super().__call__(x, t, context, **kw)
    frames, mask, _ = inputs()
    prefix_px = frames[:, :5].clone()                                                       # the previous window's last frames, as wgp.py passes them
    kw = dict(width=48, height=32, input_frames=frames, input_masks=mask, overlapped_latents=torch.zeros(1, 16, 2, 4, 6), overlap_noise=20,
              prefix_frames_count=5, return_latent_slice=slice(-2, None), window_no=2, video_prompt_type="V", model_type="vace_14B")
    outs = []
    for extra in (dict(input_video=prefix_px, prefix_video=prefix_px, pre_video_frame=prefix_px[:, -1]), dict()):
        torch.manual_seed(5)
        outs.append(run(WanAny2VHIP(VaceDiT("A"), vae=FakeVAE(), device="cpu"), **_wgp_keywords(**kw, **extra)))
    assert torch.equal(outs[0]["latents"], outs[1]["latents"]) and torch.equal(outs[0]["latent_slice"], outs[1]["latent_slice"])
    assert tuple(outs[0]["latents"].shape[-2:]) == (4, 6)


def test_progress_protocol_matches_the_references_callback_calls():
    """any2video.py:1410-1411, :1434-1436, :1442, :1446, :1743-1750: what wgp.py's callback and set_header_text receive."""
    calls, headers = [], []

    def cb(step, latents=None, force=False, override_num_inference_steps=-1, denoising_extra="", **kw):
        calls.append((step, None if latents is None else tuple(latents.shape), force, override_num_inference_steps, denoising_extra))
    a, b = FakeDiT("A"), FakeDiT("B")
    run(WanAny2VHIP(a, b, device="cpu"), guide_phases=2, switch_threshold=800, callback=cb, set_header_text=headers.append)
    n_high = len(a.calls)
    assert 0 < n_high < 6
    assert headers == [f"Denoising Steps:  Phase 1 = 1:{n_high}, Phase 2 = {n_high + 1}:6"]
    assert calls[0] == (-1, None, True, -1, "") and calls[1] == (-1, None, True, 6, "Phase 1/2 High Noise")
    body = calls[2:]
    switch = [c for c in body if c[1] is None]
    assert switch == [(n_high - 1, None, False, -1, "Phase 2/2 Low Noise")]                    # callback(step_no - 1, denoising_extra=)
    steps = [c for c in body if c[1] is not None]
    assert [c[0] for c in steps] == list(range(6)) and all(c[1] == (16, 3, 8, 8) and c[2] is False for c in steps)
    assert [c[4] for c in steps] == ["Phase 1/2 High Noise"] * n_high + ["Phase 2/2 Low Noise"] * (6 - n_high)
    assert body.index(switch[0]) == n_high                                                     # in front of the first low-noise step
    # one expert, several phases: no noise-level suffix; one phase: empty extra, no header
    calls.clear(), headers.clear()
    run(WanAny2VHIP(FakeDiT("A"), device="cpu"), guide_phases=2, switch_threshold=800, callback=cb, set_header_text=headers.append)
    assert calls[1][4] == "Phase 1/2" and calls[-1][4] == "Phase 2/2" and len(headers) == 1
    calls.clear(), headers.clear()
    run(WanAny2VHIP(FakeDiT("A"), device="cpu"), callback=cb, set_header_text=headers.append)
    assert headers == [] and all(c[4] == "" for c in calls) and calls[1][3] == 6


def test_positional_callbacks_keep_working_and_previews_leave_out_the_padded_end_frame():
    seen = []
    run(WanAny2VHIP(FakeDiT("A"), device="cpu"), guide_phases=2, switch_threshold=800, callback=lambda i, lat, force: seen.append((i, force)))
    assert seen == [(-1, True), (-1, True)] + [(i, False) for i in range(6)]
    # a callable that takes keywords but needs the latents (the GPU loop tests' tracing lambdas): no latent-less phase notice
    trace, kws = [], []
    run(WanAny2VHIP(FakeDiT("A"), FakeDiT("B"), device="cpu"), guide_phases=2, switch_threshold=800,
        callback=lambda i, l, *a, **k: (kws.append(k), trace.append(l.shape) if i >= 0 else None))
    assert len(trace) == 6 and kws[1]["override_num_inference_steps"] == 6 and kws[-1]["denoising_extra"] == "Phase 2/2 Low Noise"
    from oracle.make_golden_i2v_cond import FakeVAE
    m = FakeDiT("A")
    m.model_type = "i2v"
    shapes = []
    img = torch.rand(3, 64, 64) * 2 - 1
    out = run(WanAny2VHIP(m, vae=FakeVAE(), device="cpu"), image_start=img, image_end=img, clip_fea=torch.zeros(1, 257, 1280),
              callback=lambda i, lat, force: shapes.append(None if lat is None else lat.shape[1]))
    assert out["latents"].shape[2] == 3 and shapes[2:] == [3] * 6          # 4 latent frames inside the loop, 3 shown


def test_video_to_video_keeps_the_whole_schedule_for_cache_thresholds_and_lora_steps():
    """any2video.py:546, :1404-1406, :1444, :1493: when the schedule is cut short, the step-skipping thresholds and the LoRA
    multipliers still refer to the uncut schedule -- start step max(cache.start_step, start_step_no), step start_step_no + i."""
    from oracle.make_golden_i2v_cond import FakeVAE
    from wan2gp_amd.skipcache import SkipStepsCache
    from wan2gp_amd.lora import parse_loras_multipliers

    class Rec:
        def __init__(self):
            self.steps = []

        def set_step(self, slists, n, step_no, s1, s2):
            self.steps.append((n, step_no))
    vid = torch.rand(3, 9, 64, 64, generator=torch.Generator().manual_seed(0)) * 2 - 1
    m = FakeDiT("A")
    m.cache = SkipStepsCache(cache_type="mag", multiplier=2.0, start_step=1, magcache_K=2, magcache_thresh=0, def_mag_ratios=[0.99] * 10)
    m.loras = Rec()
    _, slists, err = parse_loras_multipliers("1", 1, 6)
    assert err == ""
    run(WanAny2VHIP(m, vae=FakeVAE(), device="cpu"), input_frames=vid, video_prompt_type="G", denoising_strength=0.5, guide_scale=1.0,
        loras_slists=slists)
    assert len(m.calls) == 3
    assert m.thresholds == ("mag", 3, 6, 2.0) and m.cache.num_steps == 6
    assert m.loras.steps == [(6, 3), (6, 4), (6, 5)]


def test_later_sliding_windows_are_colour_matched_to_their_reference_frame_after_decoding():
    """any2video.py:552, :667-687, :1008-1009, :1153-1154, :1783-1808: with color_correction_strength > 0 (its default is 1) and
    window_start_frame_no + prefix_frames_count > 1 the decoded window is Lab-matched to a reference frame -- the last prefix frame
    of the control video (VACE), the last frame of the video being continued (i2v); never with an end image, never for the first
    window, never for strength 0."""
    from oracle.make_golden_vace_context import FakeVAE, inputs
    from wan2gp_amd.color import correct_window

    class DecVAE(FakeVAE):
        def decode_to_cpu_uint8(self, zs, tile_size=0):
            out = []
            for z in zs:
                v = torch.nn.functional.interpolate(z[None, :3].float(), scale_factor=(1, 8, 8), mode="nearest")[0]
                out.append(((v * 0.6).tanh() * 0.8 + 1.0).mul(127.5).round().clamp(0, 255).to(torch.uint8))
            return out

    class VaceDiT(FakeDiT):
        vace_layers = (0,)

        def __call__(self, x, t, context, vace_context=None, vace_context_scale=None, **kw):
            return super().__call__(x, t, context, **kw)
    frames, mask, _ = inputs()
    kw = dict(width=48, height=32, input_frames=frames, input_masks=mask, return_latents=False)
    plain = run(WanAny2VHIP(VaceDiT("A"), vae=DecVAE(), device="cpu"), color_correction_strength=0, prefix_frames_count=3, window_start_frame_no=5, **kw)["x"]
    first = run(WanAny2VHIP(VaceDiT("A"), vae=DecVAE(), device="cpu"), prefix_frames_count=1, window_start_frame_no=0, **kw)["x"]
    noprefix = run(WanAny2VHIP(VaceDiT("A"), vae=DecVAE(), device="cpu"), prefix_frames_count=0, window_start_frame_no=5, **kw)["x"]
    assert plain.dtype == torch.uint8 and torch.equal(first, plain) and torch.equal(noprefix, plain)      # no reference frame without a prefix
    for strength in (1, 0.4):
        got = run(WanAny2VHIP(VaceDiT("A"), vae=DecVAE(), device="cpu"), color_correction_strength=strength, prefix_frames_count=3,
                  window_start_frame_no=5, **kw)["x"]
        assert torch.equal(got, correct_window(plain, frames[:, 2:3], strength)) and not torch.equal(got, plain)
    # VACE with a background mask on the first reference image and no prefix: that image is the colour reference (:1139)
    _, _, refs = inputs()
    bgm = (torch.rand(1, 1, 32, 48, generator=torch.Generator().manual_seed(78)) > 0.5).float()
    kb = dict(kw, input_ref_images=refs, input_ref_masks=[bgm, None], prefix_frames_count=0, window_start_frame_no=5)
    base_bg = run(WanAny2VHIP(VaceDiT("A"), vae=DecVAE(), device="cpu"), color_correction_strength=0, **kb)["x"]
    assert torch.equal(run(WanAny2VHIP(VaceDiT("A"), vae=DecVAE(), device="cpu"), **kb)["x"], correct_window(base_bg, refs[0], 1))
    # i2v: the reference frame is the last frame of the continued video; an end image switches the correction off
    g = torch.Generator().manual_seed(3)
    prefix = torch.rand(3, 5, 64, 64, generator=g) * 2 - 1

    class I2VVAE(_StubVAE, DecVAE):
        pass

    def i2v(**over):
        m = FakeDiT("A")
        m.model_type = "i2v2_2"
        return run(WanAny2VHIP(m, vae=I2VVAE(), device="cpu"), input_video=prefix, return_latents=False, **over)["x"]
    base = i2v(color_correction_strength=0, prefix_frames_count=5, window_start_frame_no=12)
    assert torch.equal(i2v(prefix_frames_count=5, window_start_frame_no=12), correct_window(base, prefix[:, -1:], 1))
    assert torch.equal(i2v(prefix_frames_count=1, window_start_frame_no=0), base)
    with_end = dict(image_end=prefix[:, 0], prefix_frames_count=5, window_start_frame_no=12)
    assert torch.equal(i2v(**with_end), i2v(color_correction_strength=0, **with_end))


def test_input_video_sets_height_and_width():
    """any2video.py:571: `if input_video is not None: height, width = input_video.shape[-2:]` -- the keywords' size is overridden."""
    m = FakeDiT("A")
    m.model_type = "i2v2_2"
    out = run(WanAny2VHIP(m, vae=_StubVAE(), device="cpu"), input_video=torch.zeros(3, 1, 32, 48), width=64, height=64)
    assert tuple(out["latents"].shape) == (1, 16, 3, 4, 6)


def test_ti2v_without_a_source_video_rounds_the_size_down_to_multiples_of_32():
    """any2video.py:1063-1065."""
    m = FakeDiT("A", out_dim=48)
    m.model_type = "ti2v2_2"
    pipe = WanAny2VHIP(m, device="cpu")
    pipe.vae_stride = (4, 16, 16)
    out = run(pipe, width=80, height=120)
    assert tuple(out["latents"].shape) == (1, 48, 3, 96 // 16, 64 // 16)


def _ref_resize_lanczos():
    """`resize_lanczos` lifted from the reference (shared/utils/utils.py:341-347) when the tree is present."""
    import ast, os
    src = os.path.join(os.environ.get("WAN_REFERENCE_ROOT", "/root/reference"), "shared", "utils", "utils.py")
    if not os.path.isfile(src):
        return None
    fn = next(n for n in ast.parse(open(src).read()).body if isinstance(n, ast.FunctionDef) and n.name == "resize_lanczos")
    import numpy as np
    from PIL import Image
    ns = {"torch": torch, "np": np, "Image": Image}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "utils.py", "exec"), ns)
    return ns["resize_lanczos"]


def test_wan21_i2v_takes_its_clip_features_from_the_host_applications_tower():
    """any2video.py:945-954: without clip_fea the pipeline asks `self.clip` (the host application's CLIPModel, handed over by the plugin's
    load_model): start image = last frame of input_video, Lanczos-resized through an 8-bit image to the tower's input size, as [3,1,S,S];
    flf2v_720p sends start AND end image (the start twice without an end image).  The resize equals the reference's own function."""
    pytest.importorskip("PIL")
    seen, got = [], []

    class Clip:
        model = types.SimpleNamespace(image_size=16)

        def visual(self, videos):
            seen.append([v.clone() for v in videos])
            return torch.full((len(videos), 257, 1280), 0.5)

    class Dit(FakeDiT):
        model_type = "i2v"

        def __call__(self, x, t, context, **kw):
            got.append(kw.get("clip_fea"))
            return super().__call__(x, t, context, **kw)
    g = torch.Generator().manual_seed(9)
    video = torch.rand(3, 5, 64, 64, generator=g) * 2 - 1
    end = torch.rand(3, 64, 64, generator=g) * 2 - 1
    from oracle.make_golden_i2v_cond import FakeVAE
    pipe = WanAny2VHIP(Dit("A"), vae=FakeVAE(), device="cpu")
    pipe.clip = Clip()
    run(pipe, input_video=video, sampling_steps=1)
    assert len(seen) == 1 and [tuple(v.shape) for v in seen[0]] == [(3, 1, 16, 16)] and tuple(got[0].shape) == (1, 257, 1280)
    ref = _ref_resize_lanczos()
    if ref is not None:
        assert torch.equal(seen[0][0][:, 0], ref(video[:, -1].clone(), 16, 16))
    assert torch.equal(seen[0][0][:, 0], pipeline._resize_lanczos(video[:, -1].clone(), 16, 16))
    # an explicit clip_fea wins; flf2v: two images
    seen.clear(), got.clear()
    cf = torch.zeros(1, 257, 1280)
    run(pipe, input_video=video, clip_fea=cf, sampling_steps=1)
    assert not seen and got[0] is cf
    pipe.flf = True
    run(pipe, input_video=video, image_end=end, sampling_steps=1)
    assert [tuple(v.shape) for v in seen[0]] == [(3, 1, 16, 16)] * 2 and tuple(got[-1].shape) == (2, 257, 1280)
    assert torch.equal(seen[0][1][:, 0], pipeline._resize_lanczos(end.clone(), 16, 16)) and not torch.equal(seen[0][0], seen[0][1])
    seen.clear()
    run(pipe, input_video=video, sampling_steps=1)
    assert len(seen[0]) == 2 and torch.equal(seen[0][0], seen[0][1])                    # no end image: the start image twice (:948)


def test_host_clip_is_built_like_the_reference_builds_it_and_absent_outside_the_host_application(monkeypatch):
    """wan_handler._host_clip: any2video.py:127-132 with configs/wan_i2v_14B.py:17-19's names; None when `models` / `shared` are not importable."""
    import sys
    from wan2gp_amd import wan_handler as H
    for name in ("models", "models.wan", "models.wan.modules", "models.wan.modules.clip", "shared", "shared.utils", "shared.utils.files_locator"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if "reference" not in p])
    assert H._host_clip("cpu") is None
    made = {}

    class CLIPModel:
        def __init__(self, **kw):
            made.update(kw)
    fl = types.ModuleType("shared.utils.files_locator")
    fl.locate_file = lambda rel: "/ckpts/" + rel
    fl.locate_folder = lambda rel: "/ckpts/" + rel
    mods = {"models": types.ModuleType("models"), "models.wan": types.ModuleType("models.wan"),
            "models.wan.modules": types.ModuleType("models.wan.modules"), "models.wan.modules.clip": types.ModuleType("models.wan.modules.clip"),
            "shared": types.ModuleType("shared"), "shared.utils": types.ModuleType("shared.utils"), "shared.utils.files_locator": fl}
    mods["models.wan.modules.clip"].CLIPModel = CLIPModel
    mods["shared.utils"].files_locator = fl
    for pkg in ("models", "models.wan", "models.wan.modules", "shared", "shared.utils"):
        mods[pkg].__path__ = []
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    clip = H._host_clip("cuda")
    assert isinstance(clip, CLIPModel) and made == {
        "dtype": torch.float16, "device": "cuda",
        "checkpoint_path": "/ckpts/xlm-roberta-large/models_clip_open-clip-xlm-roberta-large-vit-huge-14-bf16.safetensors",
        "tokenizer_path": "/ckpts/xlm-roberta-large"}


def test_image_start_and_image_end_are_read_on_the_i2v_path_only():
    """wgp.py:7765-7766 passes image_start / image_end to every model; the reference reads them inside `if i2v:` only
    (any2video.py:651-785).  A 5B generation from a start image is conditioned through input_video (timestep injection), never through
    an i2v `y`; a t2v / VACE model ignores both; an i2v model without any start image conditions on a black frame (:667-669)."""
    g = torch.Generator().manual_seed(4)
    img = torch.rand(3, 64, 64, generator=g) * 2 - 1
    for mt, out_dim in (("ti2v2_2", 16), ("t2v", 16)):
        a, b = FakeDiT("A", out_dim), FakeDiT("A", out_dim)
        a.model_type = b.model_type = mt
        kw = dict(input_video=img.unsqueeze(1)) if mt == "ti2v2_2" else {}
        got = run(WanAny2VHIP(a, vae=_StubVAE(), device="cpu"), image_start=img, image_end=img, **kw)
        want = run(WanAny2VHIP(b, vae=_StubVAE(), device="cpu"), **kw)
        assert all(c["y"] is None for c in a.calls) and torch.equal(got["latents"], want["latents"])
    m = FakeDiT("A")
    m.model_type = "i2v2_2"
    vae = _StubVAE()
    run(WanAny2VHIP(m, vae=vae, device="cpu"), sampling_steps=1)
    assert vae.seen == [(3, 9, 64, 64)] and m.calls[0]["y"] is not None
    m2, vae2 = FakeDiT("A"), _StubVAE()
    m2.model_type = "i2v2_2"
    run(WanAny2VHIP(m2, vae=vae2, device="cpu"), image_start=torch.full((3, 64, 64), -1.0), sampling_steps=1)
    assert torch.equal(m.calls[0]["y"], m2.calls[0]["y"])                               # = a start image that is -1 everywhere


def test_wgp_keyword_set_for_the_5B_model_with_a_start_image_and_for_vace_with_a_control_video():
    """The exact keyword set of wgp.py:7762-7885 for (a) ti2v_2_2 from a start image -- wgp.py passes the image BOTH as `image_start`
    [3,H,W] and as `input_video` [3,1,H,W] (wgp.py:7375-7378, :7765, :7773) -- and (b) VACE with a control video + mask and a start image
    in `image_start` (which the reference's VACE path never reads): same result as the direct call without the plumbing."""
    from oracle.make_golden_vace_context import FakeVAE, inputs
    g = torch.Generator().manual_seed(6)
    img = torch.rand(3, 64, 64, generator=g) * 2 - 1

    class Vae:
        def encode(self, videos, tile_size=0):
            return [torch.full((16, 1, 8, 8), 7.0)]
    a, b = FakeDiT("A"), FakeDiT("A")
    a.model_type = b.model_type = "ti2v2_2"
    got = run(WanAny2VHIP(a, vae=Vae(), device="cpu"),
              **_wgp_keywords(image_start=img, input_video=img.unsqueeze(1), prefix_video=img.unsqueeze(1), pre_video_frame=img,
                              conditioning_latents_size=1, perturbation_layers=None, model_type="ti2v_2_2"))
    want = run(WanAny2VHIP(b, vae=Vae(), device="cpu"), input_video=img.unsqueeze(1), shift=5.0)
    assert torch.equal(got["latents"], want["latents"]) and all(c["y"] is None for c in a.calls)

    class VaceDiT(FakeDiT):
        vace_layers = (0,)
        model_type = "t2v"

        def __call__(self, x, t, context, vace_context=None, vace_context_scale=None, **kw):
            self.scales = list(vace_context_scale)
            return super().__call__(x, t, context, **kw)
    frames, mask, _ = inputs()
    a, b = VaceDiT("A"), VaceDiT("A")
    got = run(WanAny2VHIP(a, vae=FakeVAE(), device="cpu"), width=48, height=32,
              **_wgp_keywords(image_start=torch.rand(3, 32, 48, generator=g), input_frames=frames, input_masks=mask, context_scale=[0.8],
                              video_prompt_type="PV", perturbation_layers=None, model_type="vace_14B", input_ref_images=None))
    want = run(WanAny2VHIP(b, vae=FakeVAE(), device="cpu"), width=48, height=32, input_frames=frames, input_masks=mask, context_scale=[0.8], shift=5.0)
    assert torch.equal(got["latents"], want["latents"]) and a.scales == [0.8] and all(c["y"] is None for c in a.calls)
