"""TEST INFRASTRUCTURE ONLY -- loads the *reference's own* Python modules on CPU.

Works only where ``/root/reference`` exists (the build container); it never runs on
the GPU box.  It is used by ``oracle/make_golden.py`` to (a) pin the standalone
restatement in ``oracle/wan_oracle.py`` against the reference and (b) emit the golden
fixtures committed under ``tests/golden/``.

The reference cannot be imported as a package (models/wan/__init__.py pulls the whole
family; model.py:14-27 imports mmgp/diffusers/9 variant sub-packages;
shared/attention.py:14 queries a CUDA device at import), so we pre-seed ``sys.modules``
with inert stubs for everything that is *not* arithmetic on the t2v / i2v2.2 path and
then exec the real files:

    models/wan/modules/posemb_layers.py   (RoPE tables + apply)
    models/wan/modules/model.py           (WanModel and all blocks)
    models/wan/modules/vae.py             (WanVAE_)
    shared/attention.py                   (pay_attention -> sdpa)
    shared/utils/fm_solvers_unipc.py      (FlowUniPCMultistepScheduler)
    shared/utils/euler_scheduler.py       (EulerScheduler)
"""
import functools
import importlib.util
import inspect
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models/wan/modules/model.py"))


_loaded = None


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with P (posemb_layers), M (model), V (vae), U (unipc),
    E (euler), A (shared.attention), offload (the mmgp.offload stub)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # shared/attention.py:14 runs torch.cuda.get_device_capability at import
    torch.cuda.get_device_capability = lambda *a, **k: (9, 4)

    caches = {}
    off = _mod("mmgp.offload", shared_state={},
               get_cache=lambda n: caches.setdefault(n, {}),
               clear_caches=lambda: caches.clear())
    _mod("mmgp", offload=off)

    class _Cfg(dict):
        __getattr__ = dict.get

    class ConfigMixin:
        def register_to_config(self, **kw):
            self.config.update(kw)

    def register_to_config(init):
        @functools.wraps(init)
        def inner(self, *a, **kw):
            ba = inspect.signature(init).bind(self, *a, **kw)
            ba.apply_defaults()
            self.config = _Cfg({k: v for k, v in ba.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return inner

    class ModelMixin(torch.nn.Module):
        pass

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(s, prev_sample):
            s.prev_sample = prev_sample

        def __getitem__(s, i):
            return (s.prev_sample,)[i]

    _mod("diffusers"); _mod("diffusers.models"); _mod("diffusers.schedulers")
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin,
         SchedulerOutput=SchedulerOutput, KarrasDiffusionSchedulers=[])
    _mod("diffusers.utils", deprecate=lambda *a, **k: None, is_scipy_available=lambda: True)

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    _mod("models"); _mod("models.wan"); _mod("models.wan.modules")

    def nope(*a, **k):
        raise NotImplementedError("variant path stubbed in oracle shim")

    for name, syms in {
        "models.wan.multitalk": [], "models.wan.multitalk.multitalk_utils": ["get_attn_map_with_target"],
        "models.wan.animate": [], "models.wan.animate.motion_encoder": ["Generator"],
        "models.wan.animate.face_blocks": ["FaceAdapter", "FaceEncoder"],
        "models.wan.animate.model_animate": ["after_patch_embedding"],
        "models.wan.scail": [], "models.wan.scail.model_scail": ["build_scail_pose_tokens"],
        "models.wan.scail2": ["build_scail2_pose_tokens"],
        "models.wan.steadydancer": [], "models.wan.steadydancer.small_archs": ["FactorConv3d", "PoseRefNetNoBNV3"],
        "models.wan.steadydancer.mobilenetv2_dcd": ["DYModule"],
        "models.wan.shotplan": ["inject_shotplan_tokens"],
        "models.wan.animate2": ["animate2_attention_block", "animate2_cached_attention_block"],
    }.items():
        _mod(name, **{s: nope for s in syms})

    def _load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    ns = types.SimpleNamespace()
    ns.offload = off
    ns.P = _load("models.wan.modules.posemb_layers", "models/wan/modules/posemb_layers.py")
    ns.M = _load("models.wan.modules.model", "models/wan/modules/model.py")
    ns.V = _load("ref_vae", "models/wan/modules/vae.py")
    ns.U = _load("ref_unipc", "shared/utils/fm_solvers_unipc.py")
    ns.E = _load("ref_euler", "shared/utils/euler_scheduler.py")
    ns.D = _load("ref_dpm", "shared/utils/fm_solvers.py")
    ns.FM = _load("ref_flowmatch", "shared/utils/basic_flowmatch.py")
    ns.LCM = _load("ref_lcm", "shared/utils/lcm_scheduler.py")
    import shared.attention as A  # the real file (namespace package `shared`)
    ns.A = A
    off.shared_state["_attention"] = "sdpa"
    _loaded = ns
    return ns
