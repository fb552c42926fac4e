"""-m gpu: sub-parallel temporal windows (any2video.py:1199-1387, :1724) through the HIP model.

`generate(sub_parallel_window_size=, sub_parallel_window_overlap=)` runs one joint CFG forward per overlapping latent window
(plus an anchor frame) at every step and blends the predictions.  The window bookkeeping is pinned to the reference on CPU
(tests/golden/subparallel.npz); here the SAME pipeline code is driven twice from the same start noise -- once with WanModelHIP on
the GPU, once with a stand-in model that evaluates the pinned oracle's forward on the CPU -- and the sampled latents must agree
to the forward's bf16 noise, with the forwards the windows imply (5, 6, 6 latent frames per step: anchor frames in front of
windows 2 and 3) seen at the HIP model.  Tolerance: relative L2 <= 3e-2 after 3 UniPC steps (the plain 3-step loop bar of
test_gpu_model.py is 4e-2).  (With random weights a windowed and an unwindowed run differ by only ~3e-3 -- printed, not asserted.)
"""
import pytest
import torch

from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu


class OracleDiT:
    """Stands in for WanModelHIP in the pipeline: the oracle's bf16-plan forward on the CPU."""

    def __init__(self, W, cfg):
        self.W, self.cfg, self.cache, self.loras, self.calls = W, cfg, None, None, []
        self.out_dim, self.model_type, self.device = cfg.out_dim, cfg.model_type, torch.device("cpu")

    def __call__(self, x, t, context, freqs=None, y=None, **kw):
        xs = [u.float().cpu() for u in x]
        x.clear()
        self.calls.append((xs[0].shape[2], freqs[0].shape[0]))
        ctx = [c.cpu() if c.dim() == 3 else c.cpu().unsqueeze(0) for c in context]
        out = O.dit_forward([u if u.dim() == 5 else u.unsqueeze(0) for u in xs], t.cpu(), ctx, self.W, self.cfg,
                            y=None if y is None else y.cpu(), freqs=tuple(f.cpu() for f in freqs), dtype=torch.bfloat16)
        return [o if xs[0].dim() == 5 else o[0] for o in out]


def rel(a, b):
    return ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm()).item()


def test_windows_hip_vs_oracle_through_the_same_pipeline(monkeypatch):
    from wan2gp_amd import ops
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.pipeline import WanAny2VHIP
    cfg = O.make_config("tiny")
    W = O.synth_weights(cfg, seed=4321)
    m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers,
                    in_dim=cfg.in_dim, out_dim=cfg.out_dim).load_state_dict(W)
    _, ctx, ctx_null, _ = O.synth_inputs(cfg, 2, 8, 8, seed=7)
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(1, cfg.out_dim, 11, 8, 8, generator=g)                 # frame_num 41 -> 11 latent frames
    args = dict(width=64, height=64, frame_num=41, sampling_steps=3, guide_scale=3.0, seed=1, return_latents=True)
    win = dict(sub_parallel_window_size=17, sub_parallel_window_overlap=5)      # latent windows (0,5) (3,8) (6,11)

    seen = []
    orig_forward = WanModelHIP.forward

    def spy(self, *a, **kw):
        x = kw["x"] if "x" in kw else a[0]
        seen.append(x[0].shape[-3])
        return orig_forward(self, *a, **kw)
    monkeypatch.setattr(WanModelHIP, "__call__", spy)                           # the pipeline calls the model object
    hip = WanAny2VHIP(m)
    out_win = hip.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), latents=noise, **args, **win)["latents"]
    assert seen == [5, 6, 6] * 3, seen                                          # anchor frame in front of windows 2 and 3
    out_plain = hip.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), latents=noise, **args)["latents"]
    assert tuple(out_win.shape) == (1, cfg.out_dim, 11, 8, 8) and torch.isfinite(out_win).all()

    # the same pipeline on the CPU with the oracle as the model (HIP-only tensor helpers replaced by their definitions)
    def lincomb(tensors, coefs, out=None):
        r = sum(float(c) * t_.float() for c, t_ in zip(coefs, tensors))
        return r if out is None else out.copy_(r)
    monkeypatch.setattr(ops, "lincomb", lincomb)
    monkeypatch.setattr(ops, "cfg_combine", lambda c, u, gs, out=None: u + gs * (c - u))
    ref_model = OracleDiT(W, cfg)
    ref = WanAny2VHIP(ref_model, device="cpu")
    ref_win = ref.generate(context=ctx, context_null=ctx_null, latents=noise, **args, **win)["latents"]
    assert ref_model.calls == [(5, 5 * 16), (6, 6 * 16), (6, 6 * 16)] * 3
    e = rel(out_win, ref_win)
    d = rel(out_plain, ref_win)
    print(f"[sub-parallel windows] hip vs oracle {e:.3e}; unwindowed hip run vs windowed oracle {d:.3e}")
    assert e <= 3e-2, e
