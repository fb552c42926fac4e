"""CPU: the HIP backend's `family_handler` is discovered and mapped the way the reference does it -- `map_family_handlers`
(wgp.py:2717-2735) is lifted from the reference with `ast` when /root/reference is present (build container) and restated
otherwise, then run on `wan2gp_amd.wan_handler` as a model plugin's `model_handlers` entry (docs/PLUGINS.md:37-56)."""
import ast
import importlib
import os

import pytest

REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
PATH = "wan2gp_amd.wan_handler"


def _map_family_handlers():
    ns = {"importlib": importlib, "model_handler_sources": {PATH: {"profile_roots": ["profiles"], "plugin_id": "wan2gp-hip"}},
          "model_profile_roots_by_architecture": {}, "model_plugin_ids_by_architecture": {}}
    src = os.path.join(REF, "wgp.py")
    if os.path.isfile(src):
        tree = ast.parse(open(src).read())
        fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "map_family_handlers")
        exec(compile(ast.Module(body=[fn], type_ignores=[]), src, "exec"), ns)
    else:                                                   # the same loop, restated (wgp.py:2717-2735)
        def map_family_handlers(family_handlers):
            base, infos, eqv, comp = {}, {"unknown": (100, "Unknown")}, {}, {}
            for path in family_handlers:
                handler = importlib.import_module(path).family_handler
                for model_type in handler.query_supported_types():
                    if model_type in base:
                        raise Exception(f"Model type {model_type} supported by {base[model_type].__name__} and {handler.__name__}")
                    base[model_type] = handler
                    ns["model_plugin_ids_by_architecture"][model_type] = "wan2gp-hip"
                infos.update(handler.query_family_infos())
                e, c = handler.query_family_maps()
                eqv.update(e); comp.update(c)
            return base, infos, eqv, comp
        ns["map_family_handlers"] = map_family_handlers
    return ns


def test_handler_is_mapped_like_a_builtin_family():
    ns = _map_family_handlers()
    handlers, infos, eqv, comp = ns["map_family_handlers"]([PATH])
    from wan2gp_amd.wan_handler import family_handler
    assert set(handlers) == set(family_handler.query_supported_types()) and all(h is family_handler for h in handlers.values())
    assert "t2v_2_2_hip" in handlers and "i2v_2_2_hip" in handlers and "ti2v_2_2_hip" in handlers
    assert infos["wan2_2"] == (1, "Wan2.2") and eqv["t2v_2_2_hip"] == "t2v_hip" and "vace_14B_hip" in comp["t2v_hip"]
    assert ns["model_plugin_ids_by_architecture"]["t2v_hip"] == "wan2gp-hip"
    with pytest.raises(Exception, match="supported by"):          # two handlers claiming one type (wgp.py:2727-2729)
        ns["map_family_handlers"]([PATH, PATH])


def test_model_def_properties_and_settings():
    from wan2gp_amd.wan_handler import family_handler as H
    d = H.query_model_def("t2v_2_2_hip", {"URLs2": ["x"]})
    assert d["t2v_class"] and not d["i2v_class"] and d["multiple_submodels"] and d["no_steps_skipping"] and not d["tea_cache"]
    assert d["group"] == "wan2_2" and d["profiles_dir"] == ["wan_2_2"] and d["fps"] == 16 and d["vae_block_size"] == 16
    assert [s[1] for s in d["sample_solvers"]] == ["unipc", "euler", "dpm++", "causvid", "lcm"]
    d = H.query_model_def("ti2v_2_2_hip", {})
    assert d["wan_5B_class"] and d["fps"] == 24 and d["vae_block_size"] == 32 and d["profiles_dir"] == ["wan_2_2_5B"]
    d = H.query_model_def("i2v_hip", {})
    assert d["i2v_class"] and d["black_frame"] and d["motion_amplitude"] and d["profiles_dir"] == ["wan_i2v"] and d["tea_cache"]
    # flf2v_720p (first + last frame): an i2v-class model (wan_handler.py:33, :85, :103, :424) -- start AND end image, no NAG here
    d = H.query_model_def("flf2v_720p_hip", {})
    assert d["i2v_class"] and d["black_frame"] and d["image_prompt_types_allowed"] == "SEV" and d["NAG"] is False and d["group"] == "wan"
    assert "flf2v_720p_hip" in H.query_supported_types()
    eqv, comp = H.query_family_maps()
    assert eqv["flf2v_720p_hip"] == "i2v_hip" and comp["i2v_hip"] == ["flf2v_720p_hip"]
    ui = {}
    H.update_default_settings("i2v_2_2_hip", {"image_prompt_types_allowed": "SEV"}, ui)
    assert ui == {"sample_solver": "unipc", "image_prompt_type": "S", "masking_strength": 0.1, "denoising_strength": 0.9,
                  "sliding_window_overlap": 1, "sliding_window_color_correction_strength": 0}          # wan_handler.py:1441-1449
    assert H.validate_generative_settings("t2v_hip", {}, {"sample_solver": "euler"}) is None
    assert "Unsupported" in H.validate_generative_settings("t2v_hip", {}, {"sample_solver": "ddim"})
    assert "image" in H.validate_generative_settings("t2v_hip", {}, {"sample_solver": "unipc", "image_mode": 1})
    assert H.validate_generative_settings("t2v_hip", {}, {"sample_solver": "unipc", "image_mode": 0}) is None
    assert H.query_model_family() == "wan"
    with pytest.raises(NotImplementedError):
        H.load_model(["a.safetensors"], "t2v_hip", "t2v_hip", {}, quantizeTransformer=True)
    with pytest.raises(ValueError, match="not supported"):
        H.load_model(["a.safetensors"], "multitalk", "multitalk", {}, state_dicts=[{}])


def test_model_definition_advertises_nag_and_image_prompt_types():
    """wan_handler.py:956-978, :994: the UI reads `NAG` and `image_prompt_types_allowed` (Start / End image, Video to continue,
    Last frames) from the model definition.  Only what generate() serves is claimed: no 'L' (continue the last video: a
    sliding-window feature); 'V' where the prefix-video / timestep-injection path exists."""
    from wan2gp_amd.wan_handler import family_handler as fh
    i2v = fh.query_model_def("i2v_2_2_hip", {"URLs2": ["x"]})
    assert i2v["NAG"] and i2v["image_prompt_types_allowed"] == "SEV"
    assert fh.query_model_def("t2v_2_2_hip", {})["image_prompt_types_allowed"] == "T"
    assert fh.query_model_def("ti2v_2_2_hip", {})["image_prompt_types_allowed"] == "TSV"
    assert fh.query_model_def("vace_14B_hip", {})["NAG"]
    # sliding windows: claimed where generate() carries the reference's window mechanics (i2v prefix video, 5B timestep injection,
    # VACE's pinned context overlap), not for plain t2v
    for b, want in (("t2v", False), ("t2v_2_2", False), ("t2v_1.3B", False), ("i2v", True), ("i2v_2_2", True), ("flf2v_720p", True),
                    ("ti2v_2_2", True), ("vace_14B", True), ("vace_1.3B", True)):
        assert fh.query_model_def(b + "_hip", {})["sliding_window"] is want, b


def _ref_update_default_settings():
    """wan_handler.update_default_settings (:1252-1455) lifted from the reference with `ast`, together with the module-level
    `test_*` predicates it calls (bodies untouched).  None when the reference tree is absent."""
    src = os.path.join(REF, "models", "wan", "wan_handler.py")
    if not os.path.isfile(src):
        return None
    tree = ast.parse(open(src).read())
    preds = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name.startswith("test_")]
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "family_handler")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "update_default_settings")
    fn.decorator_list = []
    # `test_scail2` is imported from models/wan/scail2/__init__.py: lift it (and the constants it reads) from there
    s2 = ast.parse(open(os.path.join(REF, "models", "wan", "scail2", "__init__.py")).read())
    s2_fn = next(n for n in s2.body if isinstance(n, ast.FunctionDef) and n.name == "test_scail2")
    used = {n.id for n in ast.walk(s2_fn) if isinstance(n, ast.Name)}
    s2_consts = [n for n in s2.body if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in used for t in n.targets)]
    ns = {}
    exec(compile(ast.Module(body=s2_consts + [s2_fn] + preds + [fn], type_ignores=[]), src, "exec"), ns)
    return ns["update_default_settings"]


@pytest.mark.parametrize("b,model_def", [("t2v", {}), ("t2v_1.3B", {}), ("t2v_2_2", {"multiple_submodels": True}),
                                         ("i2v", {}), ("flf2v_720p", {}), ("i2v_2_2", {"multiple_submodels": True}), ("ti2v_2_2", {}),
                                         ("vace_14B", {}), ("vace_1.3B", {})])
def test_default_settings_equal_the_references(b, model_def):
    """For every supported type the defaults the HIP handler writes are the ones the reference's own function writes for the
    corresponding built-in type (run on the lifted function in the build container)."""
    ref = _ref_update_default_settings()
    if ref is None:
        pytest.skip("reference tree not present")
    from wan2gp_amd.wan_handler import family_handler as H
    md = dict(H.query_model_def(b + "_hip", {"URLs2": ["x"]} if model_def.get("multiple_submodels") else {}), **model_def)
    want, got = {}, {}
    ref(b, dict(md), want)
    H.update_default_settings(b + "_hip", dict(md), got)
    assert got == want, (b, got, want)


def _ref_query_model_def():
    """wan_handler.query_model_def (:216-1007) lifted with `ast` together with every module-level function / constant it may read;
    UI-only names (gradio, file locator, prompt-info tables of other variants) are inert stand-ins."""
    src = os.path.join(REF, "models", "wan", "wan_handler.py")
    if not os.path.isfile(src):
        return None
    tree = ast.parse(open(src).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "family_handler")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "query_model_def")
    fn.decorator_list = []
    s2 = ast.parse(open(os.path.join(REF, "models", "wan", "scail2", "__init__.py")).read())
    body = ([n for n in s2.body if isinstance(n, ast.Assign)] + [n for n in s2.body if isinstance(n, ast.FunctionDef) and n.name.startswith("test_scail2")]
            + [n for n in tree.body if isinstance(n, (ast.Assign, ast.FunctionDef))] + [fn])

    class Inert:
        def __getattr__(self, k): return Inert()
        def __call__(self, *a, **k): return Inert()
        def __getitem__(self, k): return Inert()
        def __iter__(self): return iter([])
    import re as _re
    hf = {"posixpath": __import__("posixpath")}                      # shared/utils/hf.py: build_hf_url is a pure function -- the real one
    exec(compile(open(os.path.join(REF, "shared", "utils", "hf.py")).read(), "hf.py", "exec"), hf)
    ns = {"os": os, "re": _re, "gr": Inert(), "fl": Inert(), "VACE_INFOS": "", "SHOTPLAN_PROMPT_ENHANCER": "", "SHOTPLAN_PROMPT_INFOS": "",
          "get_bernini_infos": lambda *a, **k: "", "get_bernini_prompt_infos": lambda *a, **k: "",
          "get_kiwi_variant_model_def": lambda *a, **k: {}, "build_hf_url": hf["build_hf_url"]}
    exec(compile(ast.Module(body=body, type_ignores=[]), src, "exec"), ns)
    return ns["query_model_def"]


@pytest.mark.parametrize("b,md", [("t2v", {}), ("t2v_1.3B", {}), ("t2v_2_2", {"URLs2": ["x"]}), ("i2v", {}), ("i2v_2_2", {"URLs2": ["x"]}),
                                  ("ti2v_2_2", {}), ("vace_14B", {}), ("vace_1.3B", {}), ("flf2v_720p", {}),
                                  ("t2v", {"text_encoder_folder": "my-t5"}), ("i2v", {"text_encoder_URLs": ["https://h/x.safetensors"]})])
def test_model_definition_agrees_with_the_references_on_every_shared_property(b, md):
    """Every property both handlers write (class flags, fps, frame grid, VAE block size, profile folders, samplers, guidance /
    step-skipping capabilities, NAG, image prompt types, ...) has the reference's value for the corresponding built-in type --
    except `compile` (the reference names modules for torch.compile; nothing to compile here).  Properties only the reference
    writes are UI features this backend does not claim (`perturbation`, upsamplers, ...)."""
    ref = _ref_query_model_def()
    if ref is None:
        pytest.skip("reference tree not present")
    from wan2gp_amd.wan_handler import family_handler as H
    want, got = ref(b, dict(md)), H.query_model_def(b + "_hip", dict(md))
    shared = (set(want) & set(got)) - {"compile"} - ({"perturbation"} if b.startswith("vace") else set())   # (no skip-layer guidance beside VACE blocks)
    # deliberate: sliding windows are claimed only where generate() carries the window mechanics (never where the reference does not
    # claim them), and the image prompt types claimed are a subset of the reference's letters
    assert (not got["sliding_window"] or want["sliding_window"]) and set(got["image_prompt_types_allowed"]) <= set(want["image_prompt_types_allowed"])
    shared -= {"sliding_window", "image_prompt_types_allowed"}
    if b == "flf2v_720p":                                     # deliberate: NAG is not claimed beside the two-image CLIP context (refused by the driver)
        assert got["NAG"] is False and want["NAG"] is True
        shared -= {"NAG"}
    assert len(shared) >= 26
    assert {k: got[k] for k in shared} == {k: want[k] for k in shared}
    assert got.get("perturbation") == (not b.startswith("vace"))               # skip-layer guidance: claimed except beside VACE blocks


def _ref_static(name):
    """A static method of the reference's family_handler lifted with the module-level predicates (see _ref_update_default_settings)."""
    src = os.path.join(REF, "models", "wan", "wan_handler.py")
    if not os.path.isfile(src):
        return None
    tree = ast.parse(open(src).read())
    preds = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name.startswith("test_")]
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "family_handler")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []
    s2 = ast.parse(open(os.path.join(REF, "models", "wan", "scail2", "__init__.py")).read())
    s2_body = [n for n in s2.body if isinstance(n, ast.Assign)] + [n for n in s2.body if isinstance(n, ast.FunctionDef) and n.name.startswith("test_scail2")]
    ns = {"os": os}
    exec(compile(ast.Module(body=s2_body + preds + [fn], type_ignores=[]), src, "exec"), ns)
    return ns[name]


@pytest.mark.parametrize("b", ["t2v", "t2v_1.3B", "t2v_2_2", "i2v", "flf2v_720p", "i2v_2_2", "ti2v_2_2", "vace_14B", "vace_1.3B"])
def test_fix_settings_migrates_old_settings_like_the_reference(b):
    ref = _ref_static("fix_settings")
    if ref is None:
        pytest.skip("reference tree not present")
    from wan2gp_amd.wan_handler import family_handler as H
    md0 = H.query_model_def(b + "_hip", {"URLs2": ["x"]} if b.endswith("2_2") and b != "ti2v_2_2" else {})
    saved = [{}, {"sample_solver": ""}, {"sample_solver": "euler", "switch_threshold": 875}, {"guidance_phases": 2}, {"guidance_phases": 2, "image_prompt_type": "S"},
             {"image_prompt_type": "", "sliding_window_overlap": 5}]
    for version in (2.0, 2.23, 2.24, 2.3, 2.31, 2.32, 2.46, 2.47, 2.6):
        for extra in ({}, {"loras_multipliers": ["1;0;0", "0;1;0"]}, {"self_refiner": True}):
            for ui in saved:
                md = dict(md0, **extra)
                want, got = dict(ui), dict(ui)
                ref(b, version, dict(md), want)
                H.fix_settings(b + "_hip", version, dict(md), got)
                assert got == want, (b, version, extra, ui, got, want)


@pytest.mark.parametrize("b,md", [("t2v", {}), ("t2v", {"URLs2": ["x"]}), ("t2v_1.3B", {}), ("t2v_2_2", {"URLs2": ["x"]}), ("i2v", {}),
                                  ("flf2v_720p", {}), ("i2v_2_2", {"URLs2": ["x"]}), ("ti2v_2_2", {}), ("vace_14B", {}), ("vace_1.3B", {})])
def test_set_cache_parameters_hands_over_the_references_calibration_tables(b, md):
    """wgp.py:7079 calls handler.set_cache_parameters when TeaCache / MagCache is switched on: same tables, chosen the same way
    (model class; resolution for the Wan2.1 i2v model; start image + source video for the 5B model) as the lifted reference
    function, into the same attribute bag."""
    ref = _ref_static("set_cache_parameters")
    if ref is None:
        pytest.skip("reference tree not present")
    from wan2gp_amd.skipcache import SkipStepsCache
    from wan2gp_amd.wan_handler import family_handler as H
    for cache_type in ("mag", "tea"):
        for inputs in ({"resolution": "832x480"}, {"resolution": "1280x720"},
                       {"resolution": "1280x704", "image_start": object(), "video_source": "v.mp4"}):
            want, got = SkipStepsCache(), SkipStepsCache()
            ref(cache_type, b, dict(md), dict(inputs), want)
            H.set_cache_parameters(cache_type, b + "_hip", dict(md), dict(inputs), got)
            assert got.__dict__ == want.__dict__, (b, cache_type, inputs)
            assert len(getattr(got, "def_mag_ratios", getattr(got, "coefficients", []))) in (5, 78, 98)


@pytest.mark.parametrize("b", ["t2v", "t2v_1.3B", "t2v_2_2", "i2v", "flf2v_720p", "i2v_2_2", "ti2v_2_2", "vace_14B", "vace_1.3B"])
def test_lora_folders_are_the_builtin_types(b):
    """wan_handler.get_lora_dir (:150-168): same folder, same command-line overrides as the built-in type."""
    import types
    ref = _ref_static("get_lora_dir")
    if ref is None:
        pytest.skip("reference tree not present")
    from wan2gp_amd.wan_handler import family_handler as H
    for args in (types.SimpleNamespace(), types.SimpleNamespace(lora_dir="/x/t2v", lora_dir_i2v="/x/i2v"),
                 types.SimpleNamespace(lora_dir_wan="/y/wan", lora_dir_wan_i2v="/y/i2v", lora_dir_wan_1_3b="/y/1.3", lora_dir_wan_5b="/y/5")):
        assert H.get_lora_dir(b + "_hip", args, "loras") == ref(b, args, "loras"), (b, vars(args))


def test_preview_factors_come_from_the_host_applications_table(monkeypatch):
    """wan_handler.get_rgb_factors (:1009-1014): `shared.RGB_factors.get_rgb_factors("wan", type)` with the 5B types mapped to
    ti2v_2_2; (None, None) = "no preview" (wgp.py:8349-8352) outside the host application."""
    import sys, types
    from wan2gp_amd import wan_handler as W
    monkeypatch.delitem(sys.modules, "shared", raising=False)
    monkeypatch.delitem(sys.modules, "shared.RGB_factors", raising=False)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if "reference" not in p])
    assert W.family_handler.get_rgb_factors("t2v_hip") == (None, None)
    asked = []
    mod, pkg = types.ModuleType("shared.RGB_factors"), types.ModuleType("shared")
    mod.get_rgb_factors = lambda family, model_type=None, sub_family=None: (asked.append((family, model_type)), ("f", "b"))[1]
    pkg.__path__ = []
    monkeypatch.setitem(sys.modules, "shared", pkg)
    monkeypatch.setitem(sys.modules, "shared.RGB_factors", mod)
    assert W.family_handler.get_rgb_factors("ti2v_2_2_hip") == ("f", "b") and W.family_handler.get_rgb_factors("i2v_hip") == ("f", "b")
    assert asked == [("wan", "ti2v_2_2"), ("wan", "i2v")]


def test_checkpoint_files_are_resolved_by_the_host_applications_locator_when_there_is_one(monkeypatch):
    """any2video.py:143, :163: `fl.locate_file(name)` (the host's configurable checkpoint folders); outside the host, or when it does not
    find the file, `checkpoint_dir/name` -- whose absence load_model reports."""
    import sys, types
    from wan2gp_amd import wan_handler as W
    for name in ("shared", "shared.utils", "shared.utils.files_locator"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if "reference" not in p])
    assert W._locate("Wan2.1_VAE.safetensors", "ckpts") == os.path.join("ckpts", "Wan2.1_VAE.safetensors")
    fl = types.ModuleType("shared.utils.files_locator")
    fl.locate_file = lambda rel: {"Wan2.1_VAE.safetensors": "/models/wan/Wan2.1_VAE.safetensors"}.get(rel)
    pkg, sub = types.ModuleType("shared"), types.ModuleType("shared.utils")
    pkg.__path__, sub.__path__, sub.files_locator = [], [], fl
    for k, v in (("shared", pkg), ("shared.utils", sub), ("shared.utils.files_locator", fl)):
        monkeypatch.setitem(sys.modules, k, v)
    assert W._locate("Wan2.1_VAE.safetensors", "ckpts") == "/models/wan/Wan2.1_VAE.safetensors"
    assert W._locate("Wan2.2_VAE.safetensors", "ckpts") == os.path.join("ckpts", "Wan2.2_VAE.safetensors")       # not found there
    assert W._locate("https://host/repo/custom_vae.safetensors", "ckpts") == os.path.join("ckpts", "custom_vae.safetensors")
    assert W._locate("/abs/v.safetensors", "ckpts") == "/abs/v.safetensors"


def test_load_model_wires_experts_vae_text_encoder_and_the_clip_tower(monkeypatch):
    """`family_handler.load_model` end to end on stand-ins for the three device classes (the real ones need the GPU; tests/test_gpu_e2e.py
    runs it there): experts per submodel number, the VAE class per family, and for the Wan2.1 i2v class the CLIP tower handed to the host's
    offload profile under the reference's name (wan_handler.py:1156-1157) -- nothing of that for the other types."""
    import torch
    from wan2gp_amd import model as M, vae as V, vae22 as V22, wan_handler as W

    class Dit:
        def __init__(self, device=None, **arch):
            self.arch, self.model_type = arch, arch["model_type"]

        def load_state_dict(self, sd):
            self.sd = sd
            return self

    class Vae:
        def __init__(self, state_dict=None, vae_pth=None, device=None):
            self.src = ("sd", state_dict) if state_dict is not None else ("file", vae_pth)

    class Vae22(Vae):
        pass
    monkeypatch.setattr(M, "WanModelHIP", Dit)
    monkeypatch.setattr(V, "WanVAEHIP", Vae)
    monkeypatch.setattr(V22, "Wan22VAEHIP", Vae22)
    H = W.family_handler
    tower = torch.nn.Linear(2, 2)
    clip = type("Clip", (), {"model": tower})()
    te = object()
    pipe, extra = H.load_model(["hi.safetensors", "lo.safetensors"], "t2v_2_2_hip", "t2v_2_2_hip", {}, state_dicts=[{"a": 1}, {"b": 2}],
                               vae_state_dict={"v": 0}, text_encoder=te, clip=clip, profile=3, lm_decoder_engine="legacy")
    assert extra == {"pipe": {}} and pipe.clip is None and pipe.model.sd == {"a": 1} and pipe.model2.sd == {"b": 2}
    assert pipe.model.arch["mixed_precision"] is False and pipe.model2.arch["mixed_precision"] is False
    # wgp.py:4039 / :4071: the server setting "mixed_precision" arrives as mixed_precision_transformer -- both experts take the fp32-stream plan
    pmx, _ = H.load_model(["hi.safetensors", "lo.safetensors"], "t2v_2_2_hip", "t2v_2_2_hip", {}, state_dicts=[{"a": 1}, {"b": 2}],
                          vae_state_dict={"v": 0}, text_encoder=te, mixed_precision_transformer=True)
    assert pmx.model.arch["mixed_precision"] is True and pmx.model2.arch["mixed_precision"] is True
    assert type(pipe.vae) is Vae and pipe.vae.src == ("sd", {"v": 0}) and pipe.text_encoder is te and pipe.vae_stride == (4, 8, 8)
    pipe, extra = H.load_model(["m.safetensors"], "ti2v_2_2_hip", "ti2v_2_2_hip", None, state_dicts=[{}], vae_state_dict={}, text_encoder=te)
    assert type(pipe.vae) is Vae22 and pipe.vae_stride == (4, 16, 16) and pipe.model2 is None and extra == {"pipe": {}}
    for t, flf in (("i2v_hip", False), ("flf2v_720p_hip", True)):
        pipe, extra = H.load_model(["m.safetensors"], t, t, {}, state_dicts=[{}], vae_state_dict={}, text_encoder=te, clip=clip)
        assert pipe.clip is clip and pipe.flf is flf and extra == {"pipe": {"text_encoder_2": tower}}
    pipe, extra = H.load_model(["m.safetensors"], "i2v_hip", "i2v_hip", {}, state_dicts=[{}], vae_state_dict={}, text_encoder=te)
    assert pipe.clip is None and extra == {"pipe": {}}                   # test hooks + no tower given: generate() asks for clip_fea
    # a VAE named by the model definition (any2video.py:137-143)
    pipe, _ = H.load_model(["m.safetensors"], "t2v_hip", "t2v_hip", {"VAE_URLs": __file__}, state_dicts=[{}], text_encoder=te)
    assert pipe.vae.src == ("file", __file__)
    pipe, _ = H.load_model(["m.safetensors"], "t2v_hip", "t2v_hip", {"VAE_URLs": ["https://x/y/custom_vae.safetensors"]}, state_dicts=[{}],
                           text_encoder=te, checkpoint_dir="/nowhere")
    assert pipe.vae is None


@pytest.mark.parametrize("b", ["t2v", "t2v_1.3B", "t2v_2_2", "i2v", "flf2v_720p", "i2v_2_2", "ti2v_2_2", "vace_14B", "vace_1.3B"])
def test_files_to_fetch_are_a_subset_of_the_references_list(b):
    """wan_handler.query_model_files (:1016-1070; wgp.py:3659 downloads what it names): every (repository, folder, file) this handler asks
    for is one the reference's function asks for on the corresponding built-in type; the VAE, the tokenizer folder and -- Wan2.1 i2v class --
    the CLIP checkpoint are among them."""
    from wan2gp_amd.wan_handler import family_handler as H

    def triples(defs):
        return {(d["repoId"], folder, f) for d in defs for folder, fl in zip(d["sourceFolderList"], d["fileList"]) for f in fl}
    got = triples(H.query_model_files([], b + "_hip", {}))
    names = {f for _, _, f in got}
    assert ("Wan2.2_VAE.safetensors" if b == "ti2v_2_2" else "Wan2.1_VAE.safetensors") in names and "spiece.model" in names
    assert ("models_clip_open-clip-xlm-roberta-large-vit-huge-14-bf16.safetensors" in names) == (b in ("i2v", "flf2v_720p"))
    ref = _ref_static("query_model_files")
    if ref is None:
        pytest.skip("reference tree not present")
    assert got <= triples(ref([], b, {})), got - triples(ref([], b, {}))


def test_load_model_reads_quanto_int8_and_diffusers_files_the_way_wgp_picks_them(monkeypatch, tmp_path):
    """wgp.py picks the checkpoint file by the user's quantisation setting (`get_model_filename`, wgp.py:2927; int8 is the default), for
    the experts AND the text encoder (wgp.py:4051-4058): load_model runs every DiT file through the full reader (Diffusers names,
    key normalisation, quanto pairs -> bf16) and dequantises a quanto text-encoder file before the encoder sees it."""
    import torch
    from wan2gp_amd import checkpoint as C, model as M, t5 as T5, tokenizers as TK, vae as V, wan_handler as W
    files = {
        "hi_quanto_mbf16_int8.safetensors": {"model.diffusion_model.blocks.0.self_attn.q.weight._data": torch.ones(2, 2, dtype=torch.int8),
                                               "model.diffusion_model.blocks.0.self_attn.q.weight._scale": torch.ones(2, 1),
                                               "blocks.0.self_attn.q.input_scale": torch.ones(1), "head.head.bias": torch.zeros(2)},
        "t5_int8.safetensors": {"blocks.0.attn.q.weight._data": torch.ones(2, 2, dtype=torch.int8), "blocks.0.attn.q.weight._scale": torch.ones(2, 1),
                                "token_embedding.weight": torch.zeros(2, 2)},
    }
    for name in files:
        (tmp_path / name).write_bytes(b"")
    monkeypatch.setattr(C, "read_safetensors", lambda p: dict(files[os.path.basename(str(p))]))
    from wan2gp_amd import ops
    monkeypatch.setattr(ops, "dequant_i8", lambda data, scale: (data.float() * scale.view(-1, 1)).to(torch.bfloat16))
    seen = {}

    class Dit:
        def __init__(self, device=None, **arch):
            self.model_type = arch["model_type"]

        def load_state_dict(self, sd):
            seen["dit"] = sd
            return self
    monkeypatch.setattr(M, "WanModelHIP", Dit)
    monkeypatch.setattr(V, "WanVAEHIP", lambda state_dict=None, vae_pth=None, device=None: "vae")
    monkeypatch.setattr(T5, "T5EncoderModelHIP", lambda n, tok, state_dict=None, device=None: seen.setdefault("t5", state_dict))
    monkeypatch.setattr(TK, "HuggingfaceTokenizer", lambda **kw: seen.setdefault("tok", kw))
    W.family_handler.load_model([str(tmp_path / "hi_quanto_mbf16_int8.safetensors")], "t2v_hip", "t2v_hip", {}, vae_state_dict={},
                                text_encoder_filename=str(tmp_path / "t5_int8.safetensors"), device="cpu")
    assert set(seen["dit"]) == {"blocks.0.self_attn.q.weight", "head.head.bias"} and seen["dit"]["blocks.0.self_attn.q.weight"].dtype == torch.bfloat16
    assert set(seen["t5"]) == {"blocks.0.attn.q.weight", "token_embedding.weight"} and seen["t5"]["blocks.0.attn.q.weight"].dtype == torch.bfloat16
    assert seen["tok"]["name"] == str(tmp_path)                           # the tokenizer folder = the checkpoint's (any2video.py:124)
