"""Host logic of the checkpoint / LoRA row against tests/golden/loader_golden.json, which
oracle/make_golden_loader.py produced by running the REFERENCE's own functions (convert_wan.rename_key_universal,
WanModel.preprocess_sd_with_dtype, WanModel.preprocess_loras, loras_mutipliers.*).  Exact equality: keys, order, dtypes,
multiplier lists and error strings.  Plus the safetensors reader against files written by the `safetensors` package."""
import json
import os

import numpy as np
import pytest
import torch

from wan2gp_amd import checkpoint as C
from wan2gp_amd import lora as LR

with open(os.path.join(os.path.dirname(__file__), "golden", "loader_golden.json")) as _f:
    GOLD = json.load(_f)


def test_rename_diffusers_keys():
    for src, want in GOLD["rename"]:
        assert C.rename_diffusers_key(src) == want, src


@pytest.mark.parametrize("case", GOLD["preprocess_sd"], ids=lambda c: c["in"][0][0][:24])
def test_normalize_wan_keys(case):
    sd = {k: torch.zeros(2, dtype=getattr(torch, dt)) for k, dt in case["in"]}
    out = C.normalize_wan_keys(sd, torch.bfloat16)
    assert [[k, str(v.dtype).replace("torch.", "")] for k, v in out.items()] == case["out"]


@pytest.mark.parametrize("case", GOLD["preprocess_loras"], ids=lambda c: c["name"])
def test_normalize_lora_keys(case):
    sd = {k: torch.zeros(s) for k, s in case["in"]}
    vl = None if case["vace_layers"] is None else {int(a): b for a, b in case["vace_layers"].items()}
    out = LR.normalize_lora_keys(sd, case["base_model_type"], case["i2v_class"], vl)
    assert [[k, list(v.shape)] for k, v in out.items()] == case["out"]


@pytest.mark.parametrize("case", GOLD["multipliers"], ids=lambda c: (c.get("profile") or str(c["m"]))[:40].replace("\n", "/"))
def test_parse_loras_multipliers(case):
    assert LR.preparse_loras_multipliers(case["m"]) == case["preparsed"]
    first, slists, err = LR.parse_loras_multipliers(case["m"], case["n"], case["steps"], **case["kw"])
    assert err == case["error"]
    assert first == case["first"] and slists == case["slists"]
    if not err:
        s1, s2 = slists["model_switch_step"], slists["model_switch_step2"]
        exp = [LR.expand_slist(slists, i, case["steps"], s1, s2) for i in range(len(slists["phase1"]))]
        assert exp == case["expanded"]
        for step in range(case["steps"]):
            want = [e[step] if isinstance(e, list) else e for e in case["expanded"]]
            assert LR.step_multipliers(slists, case["steps"], step) == want


@pytest.mark.parametrize("case", GOLD["switch_steps"], ids=lambda c: f"g{c['in']['guide_phases']}_{c['in']['switch_threshold']}")
def test_get_model_switch_steps(case):
    assert list(LR.get_model_switch_steps(**case["in"])) == case["out"]


def test_group_adapter_and_scale():
    sd = {"diffusion_model.blocks.0.self_attn.q.lora_A.weight": torch.zeros(4, 8), "diffusion_model.blocks.0.self_attn.q.lora_B.weight": torch.zeros(8, 4),
          "diffusion_model.blocks.0.self_attn.q.alpha": torch.tensor(2.0), "blocks.1.ffn.0.lora_down.weight": torch.zeros(16, 8),
          "blocks.1.ffn.0.lora_up.weight": torch.zeros(32, 16), "blocks.1.ffn.0.diff_b": torch.zeros(32), "transformer.blocks.2.ffn.2.diff": torch.zeros(8, 32)}
    mods = LR.group_adapter(sd)
    assert set(mods) == {"blocks.0.self_attn.q", "blocks.1.ffn.0", "blocks.2.ffn.2"}
    assert LR.adapter_scale(mods["blocks.0.self_attn.q"]) == 0.5 and LR.adapter_scale(mods["blocks.1.ffn.0"]) == 1.0
    assert LR.adapter_scale(mods["blocks.2.ffn.2"]) == 1.0 and "diff_b" in mods["blocks.1.ffn.0"]
    with pytest.raises(Exception):
        LR.group_adapter({"blocks.0.ffn.0.lora_A.weight": torch.zeros(4, 8)})
    with pytest.raises(Exception):
        LR.group_adapter({"blocks.0.ffn.0.bogus": torch.zeros(1)})


def test_read_safetensors_roundtrip(tmp_path):
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(0)
    sd = {"a.weight": torch.randn(5, 7, generator=g).to(torch.bfloat16), "b": torch.randn(3, generator=g), "c.h": torch.randn(2, 2, 2, generator=g).half(),
          "q._data": torch.randint(-128, 128, (4, 6), generator=g, dtype=torch.int8), "e": torch.zeros(0, 3), "s": torch.tensor(2.5),
          "f8": torch.randn(4, 4, generator=g).to(torch.float8_e4m3fn), "i": torch.arange(6).view(2, 3)}
    p = str(tmp_path / "x.safetensors")
    save_file(sd, p, metadata={"format": "pt"})
    got, meta = C.read_safetensors(p, with_metadata=True)
    assert meta == {"format": "pt"} and set(got) == set(sd)
    for k, v in sd.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        assert torch.equal(got[k].view(torch.uint8) if v.dtype == torch.float8_e4m3fn else got[k], v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v)


def test_read_safetensors_rejects_garbage(tmp_path):
    from wan2gp_amd.lib import WanHipError
    p = tmp_path / "bad.safetensors"
    p.write_bytes(b"\x01\x02")
    with pytest.raises(WanHipError):
        C.read_safetensors(str(p))
    p.write_bytes((10 ** 9).to_bytes(8, "little") + b"{}")
    with pytest.raises(WanHipError):
        C.read_safetensors(str(p))
    hdr = json.dumps({"w": {"dtype": "F32", "shape": [4], "data_offsets": [0, 8]}}).encode()
    p.write_bytes(len(hdr).to_bytes(8, "little") + hdr + b"\0" * 8)
    with pytest.raises(WanHipError):
        C.read_safetensors(str(p))


def test_convert_diffusers_state_dict_casts():
    sd = {"blocks.0.attn1.to_q.weight": torch.zeros(2, 2), "proj_out.bias": torch.zeros(2)}
    out = C.convert_diffusers_state_dict(sd, "bf16")
    assert list(out) == ["blocks.0.self_attn.q.weight", "head.head.bias"] and all(v.dtype == torch.bfloat16 for v in out.values())
    with pytest.raises(ValueError):
        C.convert_diffusers_state_dict(sd, "int4")
