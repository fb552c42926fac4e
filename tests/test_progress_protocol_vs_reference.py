"""The callback / header-text traffic of `WanAny2VHIP.generate` against the REFERENCE's own statements: the block of
models/wan/any2video.py from `denoising_extra = ""` to `callback(-1, None, True, override_num_inference_steps = ...)` (:1430-1446,
`update_guidance` included) is lifted verbatim, driven over the same timesteps the way the loop drives it (:1491-1492, :1749),
and the recorded calls are compared with what a generate() run over a recording fake model hands its callback.  Runs where the
reference tree is present; the fixed expectations of tests/test_pipeline_control_flow_cpu.py travel.  CPU only."""
import os
import sys
import textwrap
import types

import pytest
import torch

from wan2gp_amd import lora, ops
from wan2gp_amd.pipeline import WanAny2VHIP
from tests.test_pipeline_control_flow_cpu import FakeDiT, run

REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models", "wan")), reason="reference tree not present")


@pytest.fixture(autouse=True)
def torch_stubs(monkeypatch):
    def lincomb(tensors, coefs, out=None):
        r = sum(float(c) * t_.float() for c, t_ in zip(coefs, tensors))
        return r if out is None else out.copy_(r)
    monkeypatch.setattr(ops, "lincomb", lincomb)
    monkeypatch.setattr(ops, "cfg_combine", lambda c, u, g, out=None: u + g * (c - u))
    yield


def lifted_begin():
    lines = open(os.path.join(REF, "models/wan/any2video.py")).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.strip() == 'denoising_extra = ""')
    b = next(i for i, l in enumerate(lines) if i > a and l.strip().startswith("callback(-1, None, True, override_num_inference_steps"))
    block = textwrap.dedent("\n".join(lines[a:b + 1]))
    code = ("def begin(self, callback, set_header_text, original_timesteps, updated_num_steps, guide_phases, model_switch_phase, "
            "switch_threshold, switch2_threshold, loras_slists):\n" + textwrap.indent(block, "    ") +
            "\n    return update_guidance, denoising_extra\n")
    mod = types.ModuleType("shared.utils.loras_mutipliers")                  # the two names the block imports
    mod.get_model_switch_steps = lora.get_model_switch_steps                 # (pinned to the reference in test_loader_host_vs_golden)
    mod.update_loras_slists = lambda *a, **k: None
    saved = {k: sys.modules.get(k) for k in ("shared", "shared.utils", "shared.utils.loras_mutipliers")}
    sys.modules.setdefault("shared", types.ModuleType("shared"))
    sys.modules.setdefault("shared.utils", types.ModuleType("shared.utils"))
    sys.modules["shared.utils.loras_mutipliers"] = mod
    ns = {}
    exec(compile(code, "any2video_progress_lifted.py", "exec"), ns)
    return ns["begin"], saved


def reference_traffic(timesteps, guide_phases, two_experts, model_switch_phase, th1, th2):
    begin, saved = lifted_begin()
    calls, headers = [], []

    def cb(step=-1, latent=None, force=True, override_num_inference_steps=-1, denoising_extra=""):
        calls.append((step, latent is not None, force, override_num_inference_steps, denoising_extra))
    try:
        me = types.SimpleNamespace(model="hi", model2="lo" if two_experts else None)
        cb(-1, None, True)                                                              # :1410-1411
        update_guidance, extra = begin(me, cb, headers.append, timesteps, len(timesteps), guide_phases, model_switch_phase, th1, th2, None)
        trans, g, d1, d2 = me.model, 1.0, False, False
        for i, t in enumerate(timesteps):
            g, d1, trans, extra = update_guidance(i, t, g, 2.0, d1, th1, trans, 2, extra)   # :1491
            g, d2, trans, extra = update_guidance(i, t, g, 3.0, d2, th2, trans, 3, extra)   # :1492
            cb(i, object(), False, denoising_extra=extra)                                   # :1749
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return calls, headers


@pytest.mark.parametrize("guide_phases,two,msp,th1,th2", [
    (1, False, 1, 0, 0), (2, True, 1, 800, 0), (2, False, 1, 800, 0), (2, True, 1, 1000, 0), (2, True, 1, 5, 0),
    (3, True, 2, 900, 500), (3, True, 1, 900, 500), (3, False, 1, 700, 700), (3, True, 2, 900, 1), (2, True, 2, 600, 0)])
def test_generate_talks_to_its_callback_like_the_reference(guide_phases, two, msp, th1, th2):
    ours, headers = [], []

    def cb(step=-1, latent=None, force=True, override_num_inference_steps=-1, denoising_extra=""):
        ours.append((step, latent is not None, force, override_num_inference_steps, denoising_extra))
    a, b = FakeDiT("A"), (FakeDiT("B") if two else None)
    run(WanAny2VHIP(a, b, device="cpu"), guide_phases=guide_phases, switch_threshold=th1, switch2_threshold=th2, model_switch_phase=msp,
        guide2_scale=2.0, guide3_scale=3.0, callback=cb, set_header_text=headers.append)
    ts = [torch.tensor(c["t"]) for c in sorted(a.calls + (b.calls if b else []), key=lambda c: c["step"])]
    want, want_headers = reference_traffic(ts, guide_phases, two, msp, th1, th2)
    assert ours == want
    assert headers == want_headers
