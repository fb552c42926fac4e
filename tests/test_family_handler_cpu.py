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
    ui = {}
    H.update_default_settings("i2v_2_2_hip", {"image_prompt_types_allowed": "SEV"}, ui)
    assert ui == {"sample_solver": "unipc", "image_prompt_type": "S"}
    assert H.validate_generative_settings("t2v_hip", {}, {"sample_solver": "euler"}) is None
    assert "Unsupported" in H.validate_generative_settings("t2v_hip", {}, {"sample_solver": "ddim"})
    assert H.query_model_family() == "wan" and H.query_model_files(None, "t2v_hip") == []
    with pytest.raises(NotImplementedError):
        H.load_model(["a.safetensors"], "t2v_hip", "t2v_hip", {}, quantizeTransformer=True)
    with pytest.raises(ValueError, match="not supported"):
        H.load_model(["a.safetensors"], "multitalk", "multitalk", {}, state_dicts=[{}])


def test_model_definition_advertises_nag_and_image_prompt_types():
    """wan_handler.py:956-978, :994: the UI reads `NAG` and `image_prompt_types_allowed` (Start / End image, Video to continue,
    Last frames) from the model definition."""
    from wan2gp_amd.wan_handler import family_handler as fh
    i2v = fh.query_model_def("i2v_2_2_hip", {"URLs2": ["x"]})
    assert i2v["NAG"] and i2v["image_prompt_types_allowed"] == "SEVL"
    assert fh.query_model_def("t2v_2_2_hip", {})["image_prompt_types_allowed"] == "TVL"
    assert fh.query_model_def("ti2v_2_2_hip", {})["image_prompt_types_allowed"] == "TSVL"
    assert fh.query_model_def("vace_14B_hip", {})["NAG"]
