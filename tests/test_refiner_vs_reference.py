"""Self-refining sampler steps (`wan2gp_amd/refiner.py`) against the reference's own handler: shared/utils/self_refiner.py is
loaded from the reference tree (its one third-party import, diffusers' `randn_tensor`, is a two-line stand-in) and both are driven
through the same sampler steps -- same scheduler class, same seeded noise, same stand-in model -- CPU, exact equality.  Plan
parsing is compared on a list of well- and ill-formed strings.  Skipped where the reference tree is absent."""
import copy
import importlib.util
import os
import sys
import types

import pytest
import torch

from wan2gp_amd import ops, refiner, schedulers

REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF, "shared", "utils", "self_refiner.py")
pytestmark = pytest.mark.skipif(not os.path.isfile(SRC), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    stub = types.ModuleType("diffusers.utils.torch_utils")
    stub.randn_tensor = lambda shape, generator=None, device=None, dtype=None, layout=None: torch.randn(shape, generator=generator, device=device, dtype=dtype)
    saved = {k: sys.modules.get(k) for k in ("diffusers", "diffusers.utils", "diffusers.utils.torch_utils")}
    sys.modules.setdefault("diffusers", types.ModuleType("diffusers"))
    sys.modules.setdefault("diffusers.utils", types.ModuleType("diffusers.utils"))
    sys.modules["diffusers.utils.torch_utils"] = stub
    try:
        spec = importlib.util.spec_from_file_location("ref_self_refiner", SRC)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m


@pytest.fixture(autouse=True)
def torch_lincomb(monkeypatch):
    monkeypatch.setattr(ops, "lincomb", lambda ts, cs, out=None: sum(float(c) * t_.float() for c, t_ in zip(cs, ts)))
    yield


@pytest.mark.parametrize("text", [None, "", "1-5:3", "1-5:3,6-13:1", " 2 : 4 , 7-9:2 ", "3-1:2", "1-5:3;2-4:2", "1-5", "a-3:2", "1-5:x", "1-5:", [{"start": 1, "end": 2, "steps": 3}, {"bad": 1}]])
def test_plan_parsing(ref, text):
    assert refiner.parse_plan(text) == ref.normalize_self_refiner_plan(text)


def model(x):                                                             # a stand-in denoiser: smooth, input-dependent
    return 0.3 * torch.roll(x, 1, dims=-1) - 0.2 * x + 0.05 * torch.sin(3 * x)


@pytest.mark.parametrize("solver,plan,f_unc,p", [("unipc", "1-3:3,4:2", 0.012, 1), ("unipc", "", 0.02, 2), ("euler", "0-5:4", 0.004, 1),
                                                  ("dpm++", "1-2:3", 0.0, 1), ("unipc", "1-4:3", 10.0, 1)])
def test_refined_trajectory_equals_the_references(ref, solver, plan, f_unc, p):
    def sched():
        if solver == "unipc":
            s = schedulers.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            s.set_timesteps(6, device="cpu", shift=5.0)
        elif solver == "euler":
            s = schedulers.EulerScheduler(num_train_timesteps=1000, use_timestep_transform=True)
            s.set_timesteps(6, device="cpu", shift=5.0)
        else:
            s = schedulers.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            schedulers.retrieve_timesteps(s, device="cpu", sigmas=schedulers.get_sampling_sigmas(6, 5.0))
        return s
    target_shape = (16, 3, 4, 6)
    x0 = torch.randn(1, *target_shape, generator=torch.Generator().manual_seed(4))
    outs, calls = [], []
    for impl in ("ref", "ours"):
        s, g = sched(), torch.Generator().manual_seed(11)
        h = ref.create_self_refiner_handler(plan, f_unc, p, 0.9) if impl == "ref" else refiner.create(plan, f_unc, p, 0.9)
        x, n, traj = x0.clone(), 0, []

        def denoise(z):
            nonlocal n
            n += 1
            return model(z)
        for i, t in enumerate(s.timesteps):
            pred = denoise(x)
            x, s = h.step(i, x, pred, t, s.timesteps, target_shape, g, s, {"generator": g}, denoise)
            traj.append(x.clone())
        outs.append(traj)
        calls.append(n)
    assert calls[0] == calls[1]
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    if f_unc >= 10.0:                                                     # everything settles at the first repetition: no extra model calls
        assert calls[0] == 6 + 4
    elif plan:
        assert calls[0] > 6


def test_an_interrupted_model_call_propagates_none(ref):
    s = schedulers.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    s.set_timesteps(4, device="cpu", shift=5.0)
    h = refiner.create("0-3:3", 0.0, 1, 0.999)
    x = torch.randn(1, 16, 2, 4, 4)
    out, s2 = h.step(0, x, model(x), s.timesteps[0], s.timesteps, (16, 2, 4, 4), torch.Generator().manual_seed(0), s, {}, lambda z: None)
    assert out is None
    assert h.step(0, x, None, s.timesteps[0], s.timesteps, (16, 2, 4, 4), None, s, {}, model)[0] is None
