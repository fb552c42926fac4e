"""TEST INFRASTRUCTURE ONLY -- tests/golden/loader_golden.json from the REFERENCE's own key-map / multiplier code.

Run in the build container (needs /root/reference):   python oracle/make_golden_loader.py
Executes, unmodified:
  * models/wan/convert_wan.py        rename_key_universal                       (Diffusers -> Wan names)
  * models/wan/modules/model.py      WanModel.preprocess_sd_with_dtype, WanModel.preprocess_loras
  * shared/utils/loras_mutipliers.py preparse_loras_multipliers, parse_loras_multipliers, expand_slist,
                                     get_model_switch_steps
on the synthetic key lists / multiplier strings below (tensor values never matter for these functions; tensors are
recorded by dtype + shape), plus every `loras_multipliers` string of the reference's own profiles/wan_2_2/*.json.
"""
import glob
import importlib.util
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

REF_ROOT = ref_shim.REF_ROOT
OUT = os.path.join(ROOT, "tests", "golden", "loader_golden.json")


def diffusers_keys():
    ks = []
    for i in (0, 7, 39):
        b = f"blocks.{i}."
        for a in ("attn1", "attn2"):
            for p in ("to_q", "to_k", "to_v", "to_out.0"):
                ks += [f"{b}{a}.{p}.weight", f"{b}{a}.{p}.bias"]
            ks += [f"{b}{a}.norm_q.weight", f"{b}{a}.norm_k.weight"]
        ks += [f"{b}attn2.add_k_proj.weight", f"{b}attn2.add_v_proj.bias", f"{b}attn2.norm_added_k.weight",
               f"{b}ffn.net.0.proj.weight", f"{b}ffn.net.0.proj.bias", f"{b}ffn.net.2.weight", f"{b}ffn.net.2.bias",
               f"{b}scale_shift_table", f"{b}norm2.weight", f"{b}norm2.bias"]
    ks += ["condition_embedder.text_embedder.linear_1.weight", "condition_embedder.text_embedder.linear_2.bias",
           "condition_embedder.time_embedder.linear_1.weight", "condition_embedder.time_embedder.linear_2.weight",
           "condition_embedder.time_proj.weight", "condition_embedder.time_proj.bias",
           "condition_embedder.image_embedder.norm1.weight", "condition_embedder.image_embedder.ff.net.0.proj.weight",
           "condition_embedder.image_embedder.ff.net.2.bias", "condition_embedder.image_embedder.norm2.bias",
           "proj_out.weight", "proj_out.bias", "scale_shift_table", "patch_embedding.weight", "patch_embedding.bias",
           # canonical names must pass through unchanged
           "blocks.3.self_attn.q.weight", "blocks.3.cross_attn.norm_k.weight", "blocks.3.ffn.0.weight", "head.head.weight",
           "head.modulation", "blocks.3.modulation", "blocks.3.norm3.weight", "time_projection.1.weight"]
    return ks


def wan_sd_cases():
    """(key, dtype) lists for preprocess_sd_with_dtype."""
    f8, bf = "float8_e4m3fn", "bfloat16"
    return [
        [("model.diffusion_model.blocks.0.self_attn.q.weight", bf), ("model.diffusion_model.blocks.0.self_attn.norm_q.weight", f8),
         ("model.diffusion_model.blocks.0.norm3.bias", "float8_e5m2"), ("model.diffusion_model.blocks.0.ffn.0.weight", f8),
         ("model.diffusion_model.head.head.weight", bf), ("vae.decoder.conv1.weight", bf),
         ("model.diffusion_model.blocks.1.attn2.norm_added_q.weight", bf)],
        [("blocks.0.block.self_attn.q.weight", bf), ("blocks.12.block.ffn.2.bias", bf), ("blocks.2.self_attn.k.weight", bf),
         ("patch_embedding_pose.weight", bf), ("patch_embedding_mask.bias", bf), ("patch_embedding.weight", "float32"),
         ("blocks.5.cross_attn.norm_k.weight", f8), ("blocks.5.cross_attn.k.weight", f8), ("vae.x", bf), ("blocks.block.x", bf)],
    ]


def lora_cases():
    """(name, base_model_type, i2v_class, vace_layers, [(key, shape)])."""
    r = 4
    def pair(prefix, n=8, k=8):
        return [(prefix + ".lora_A.weight", [r, k]), (prefix + ".lora_B.weight", [n, r])]
    cases = []
    ks = []
    for m in ("self_attn.q", "self_attn.o", "cross_attn.k", "cross_attn.k_img", "cross_attn.v_img", "ffn.0", "ffn.2"):
        ks += pair(f"diffusion_model.blocks.0.{m}") + [(f"diffusion_model.blocks.0.{m}.alpha", [])]
    ks += pair("diffusion_model.img_emb.proj.1") + [("diffusion_model.blocks.0.modulation.diff", [1, 6, 8]),
                                                      ("diffusion_model.head.modulation.diff", [1, 2, 8]),
                                                      ("diffusion_model.blocks.0.self_attn.norm_q.diff", [8]),
                                                      ("diffusion_model.blocks.0.self_attn.q.diff_b", [8])]
    cases.append(("diffusers_t2v", "t2v", False, None, ks))
    cases.append(("diffusers_i2v", "i2v", True, None, ks))
    cases.append(("diffusers_i2v_2_2", "i2v_2_2", True, None, ks))
    kohya = []
    for i in (0, 11):
        for m in ("self_attn_q", "self_attn_k", "cross_attn_v", "cross_attn_o", "ffn_0", "ffn_2", "cross_attn_k_img"):
            kohya += [(f"lora_unet_blocks_{i}_{m}.lora_down.weight", [r, 8]), (f"lora_unet_blocks_{i}_{m}.lora_up.weight", [8, r]),
                      (f"lora_unet_blocks_{i}_{m}.alpha", [])]
    kohya += [("lora_unet_head_head.lora_down.weight", [r, 8]), ("lora_unet_head_head.lora_up.weight", [8, r]),
              ("lora_unet_text_embedding_0.lora_down.weight", [r, 8]), ("lora_unet_text_embedding_0.lora_up.weight", [8, r]),
              ("lora_unet_time_embedding_2.lora_down.weight", [r, 8]), ("lora_unet_time_projection_1.lora_up.weight", [8, r]),
              ("lora_unet_img_emb_proj_1.lora_down.weight", [r, 8])]
    cases.append(("kohya_t2v", "t2v", False, None, kohya))
    cases.append(("kohya2_t2v", "t2v_2_2", False, None, [(k.replace("lora_unet_", "lora_unet__"), s) for k, s in kohya]))
    peft = [(k.replace(".lora_A.", ".lora_A.default.").replace(".lora_B.", ".lora_B.default."), s)
            for k, s in pair("diffusion_model.blocks.2.self_attn.v") + pair("diffusion_model.blocks.2.ffn.0")]
    cases.append(("peft_default", "t2v", False, None, peft))
    vace = []
    for i in (0, 1, 3):
        vace += pair(f"vace_blocks.{i}.self_attn.q") + pair(f"vace_blocks.{i}.after_proj")
    vace += pair("blocks.4.self_attn.q")
    cases.append(("vace", "vace_14B", False, {0: 0, 1: 5, 2: 10, 3: 15}, vace))
    cases.append(("scail", "scail", True, None, [("diffusion_model.patch_embedding.diff", [8, 20, 1, 2, 2]),
                                                  ("diffusion_model.patch_embedding.diff_b", [8])] + pair("diffusion_model.blocks.0.ffn.0")))
    cases.append(("scail2", "scail2_14B", True, None, [("diffusion_model.patch_embedding.diff", [8, 20, 1, 2, 2]),
                                                        ("diffusion_model.pose_patch_embedding.diff", [8, 16, 1, 2, 2]),
                                                        ("diffusion_model.blocks.0.cross_attn.k_img.diff", [8, 8])]))
    cases.append(("empty", "t2v", False, None, []))
    return cases


def multiplier_cases():
    cases = [
        dict(m="1;0 0;1", n=2, steps=4, kw=dict(nb_phases=2, model_switch_step=2)),
        dict(m="1;0 0;1", n=2, steps=8, kw=dict(nb_phases=3, model_switch_step=2, model_switch_step2=5, model_switch_phase=1)),
        dict(m="1;0 0;1", n=2, steps=8, kw=dict(nb_phases=3, model_switch_step=2, model_switch_step2=5, model_switch_phase=2)),
        dict(m="0.8 1.2", n=2, steps=6, kw={}),
        dict(m="0.9,0.8,0.7 1", n=2, steps=7, kw={}),
        dict(m="1,0.5;0.2,0.1,0 2", n=3, steps=10, kw=dict(nb_phases=2, model_switch_step=4)),
        dict(m="# comment line\n1.5\n0.5;0.25\n", n=2, steps=5, kw=dict(model_switch_step=3)),
        dict(m="1|2 3", n=3, steps=4, kw={}),
        dict(m="1|2|3", n=3, steps=4, kw={}),
        dict(m="1;2;3;4", n=1, steps=4, kw=dict(nb_phases=3)),
        dict(m="abc", n=1, steps=4, kw={}),
        dict(m="1,x", n=1, steps=4, kw={}),
        dict(m="", n=2, steps=4, kw={}),
        dict(m="1 2 3 4", n=2, steps=4, kw={}),
        dict(m=[1.5, "0.5;1"], n=2, steps=4, kw=dict(model_switch_step=1)),
        dict(m="1:2;3:4 5", n=2, steps=4, kw=dict(lora_multiplier_branches=["cond", "uncond"], model_switch_step=2)),
        dict(m="1:2:3", n=1, steps=4, kw=dict(lora_multiplier_branches=["cond", "uncond"])),
        dict(m="0.5", n=1, steps=0, kw={}),
        dict(m="1,2,3;4,5", n=1, steps=6, kw=dict(model_switch_step=6)),
        dict(m="1,2,3;4,5", n=1, steps=6, kw=dict(model_switch_step=0)),
    ]
    for f in sorted(glob.glob(os.path.join(REF_ROOT, "profiles", "wan_2_2", "*.json"))):
        with open(f) as fh:
            prof = json.load(fh)
        mult = prof.get("loras_multipliers")
        if isinstance(mult, (str, list)) and mult:
            steps = int(prof.get("num_inference_steps", 4))
            phases = int(prof.get("guidance_phases", 1))
            cases.append(dict(m=mult, n=len(prof.get("activated_loras", [])), steps=steps, profile=os.path.basename(f),
                              kw=dict(nb_phases=phases, model_switch_step=steps // 2, model_switch_step2=steps * 3 // 4,
                                      model_switch_phase=int(prof.get("model_switch_phase", 1)))))
    return cases


def switch_cases():
    ts = [999.0, 967.5, 921.3, 876.0, 801.2, 650.0, 421.9, 130.4]
    return [dict(timesteps=ts, guide_phases=g, model_switch_phase=1, switch_threshold=a, switch2_threshold=b)
            for g, a, b in ((1, 876, 0), (2, 876, 0), (2, 1000, 0), (2, 0, 0), (3, 900, 500), (3, 900, 900), (3, 100, 50), (3, 1000, 1000))]


def main():
    ns = ref_shim.load()
    wgp = types.ModuleType("wgp")
    state = {"i2v": False}
    wgp.test_class_i2v = lambda base: state["i2v"]
    sys.modules["wgp"] = wgp

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    conv = load("ref_convert_wan", "models/wan/convert_wan.py")
    mults = load("ref_loras_multipliers", "shared/utils/loras_mutipliers.py")
    WanModel = ns.M.WanModel

    gold = {"rename": [[k, conv.rename_key_universal(k)] for k in diffusers_keys()], "preprocess_sd": [], "preprocess_loras": [],
            "multipliers": [], "switch_steps": []}
    for case in wan_sd_cases():
        sd = {k: torch.zeros(2, dtype=getattr(torch, dt)) for k, dt in case}
        out = WanModel.preprocess_sd_with_dtype(torch.bfloat16, sd)
        gold["preprocess_sd"].append({"in": case, "out": [[k, str(v.dtype).replace("torch.", "")] for k, v in out.items()]})
    for name, base, i2v, vace_layers, keys in lora_cases():
        state["i2v"] = i2v
        sd = {k: torch.zeros(s) for k, s in keys}
        fake_self = types.SimpleNamespace(vace_layers=vace_layers)
        out = WanModel.preprocess_loras(fake_self, base, sd)
        gold["preprocess_loras"].append({"name": name, "base_model_type": base, "i2v_class": i2v,
                                         "vace_layers": None if vace_layers is None else {str(a): b for a, b in vace_layers.items()},
                                         "in": keys, "out": [[k, list(v.shape)] for k, v in out.items()]})
    for c in multiplier_cases():
        first, slists, err = mults.parse_loras_multipliers(c["m"], c["n"], c["steps"], **c["kw"])
        rec = dict(c)
        rec["first"], rec["slists"], rec["error"] = first, slists, err
        rec["preparsed"] = mults.preparse_loras_multipliers(c["m"])
        if not err:
            s1, s2 = slists["model_switch_step"], slists["model_switch_step2"]
            rec["expanded"] = [mults.expand_slist(slists, i, c["steps"], s1, s2) for i in range(len(slists["phase1"]))]
        gold["multipliers"].append(rec)
    for c in switch_cases():
        gold["switch_steps"].append({"in": c, "out": list(mults.get_model_switch_steps(**c))})
    with open(OUT, "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", OUT, {k: len(v) for k, v in gold.items()})


if __name__ == "__main__":
    main()
