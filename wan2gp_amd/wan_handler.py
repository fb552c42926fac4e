"""`family_handler` of the HIP backend -- the model-family plugin surface of the reference (seam B4).

The reference discovers model families through modules that expose a `family_handler` object: built-in ones are listed in
`family_handlers` (wgp.py:2469) and mapped by `map_family_handlers` (wgp.py:2717-2735); a *model plugin* adds more through
`plugin_info.json: {"type": "model", "model_handlers": [...]}` (docs/PLUGINS.md:37-56).  This module is such a handler: the
same static interface as `models/wan/wan_handler.py:72` for the model types the HIP path implements, whose `load_model`
(:1116-1158) builds a resident-weights `WanAny2VHIP` pipeline instead of `WanAny2V` + mmgp offload.

    plugin_info.json:  {"name": "Wan on MI355X (HIP)", "type": "model", "model_handlers": ["wan2gp_amd.wan_handler"]}

The HIP handler claims DISTINCT type names (suffix `_hip`) so that it can be installed next to the built-in Wan handler:
`map_family_handlers` raises when two handlers claim one model type (wgp.py:2727-2729).  A model definition JSON selects
it with `"architecture": "t2v_2_2_hip"` etc.  `load_model` returns `(pipeline, {"pipe": {...}})`: the modules handed to
`offload.profile` (wgp.py:4074-4090) are empty because nothing is offloaded -- all weights stay in the 288 GB of HBM.
"""
import os

import torch

# architecture of each supported base type (models/wan/configs/<type>.json of the reference)
_ARCH = {
    "t2v": dict(model_type="t2v", dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=16, out_dim=16),
    "t2v_1.3B": dict(model_type="t2v", dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, in_dim=16, out_dim=16),
    "t2v_2_2": dict(model_type="t2v", dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=16, out_dim=16),
    "i2v": dict(model_type="i2v", dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=36, out_dim=16),
    # flf2v_720p (first + last frame, models/wan/configs/flf2v_720p.json): the i2v architecture; its checkpoint carries img_emb.emb_pos
    # and the model takes the CLIP features of BOTH images (model.py:878-887, any2video.py:949-950)
    "flf2v_720p": dict(model_type="i2v", dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=36, out_dim=16),
    "i2v_2_2": dict(model_type="i2v2_2", dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=36, out_dim=16),
    "ti2v_2_2": dict(model_type="ti2v2_2", dim=3072, ffn_dim=14336, num_heads=24, num_layers=30, in_dim=48, out_dim=48),
    "vace_14B": dict(model_type="t2v", dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=16, out_dim=16,
                     vace_layers=list(range(0, 40, 5)), vace_in_dim=96),
    "vace_1.3B": dict(model_type="t2v", dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, in_dim=16, out_dim=16,
                      vace_layers=list(range(0, 30, 2)), vace_in_dim=96),
}
SUFFIX = "_hip"


def _locate(name, checkpoint_dir):
    """`fl.locate_file(name)` of the host application (shared/utils/files_locator.py: searches its configured checkpoint folders) when this
    package runs inside it, else `checkpoint_dir/name`."""
    try:
        from shared.utils import files_locator as fl
        found = fl.locate_file(name)
        if found:
            return found
    except Exception:      # noqa: BLE001 -- not inside the host application, or the file is not there: the caller checks the path it gets
        pass
    if name.startswith("http"):                          # files_locator.py:211-212: a URL stands for its file name
        name = os.path.basename(name)
    return name if os.path.isabs(name) else os.path.join(checkpoint_dir, name)


def _host_clip(device):
    """The host application's CLIP model, constructed like any2video.py:127-132 (configs/wan_i2v_14B.py:17-19).  None when this package
    is used outside Wan2GP (its `models` / `shared` packages are not importable)."""
    try:
        from models.wan.modules.clip import CLIPModel
        from shared.utils import files_locator as fl
    except ImportError:
        return None
    return CLIPModel(dtype=torch.float16, device=device,
                     checkpoint_path=fl.locate_file("xlm-roberta-large/models_clip_open-clip-xlm-roberta-large-vit-huge-14-bf16.safetensors"),
                     tokenizer_path=fl.locate_folder("xlm-roberta-large"))


def base_of(model_type: str) -> str:
    return model_type[: -len(SUFFIX)] if model_type.endswith(SUFFIX) else model_type


def test_class_i2v(base_model_type):            # wan_handler.py:16-17 restricted to the supported types
    return base_of(base_model_type) in ("i2v", "i2v_2_2", "flf2v_720p")


def test_class_t2v(base_model_type):
    return base_of(base_model_type) in ("t2v", "t2v_2_2")       # wan_handler.py:35-36: the 1.3B model is not in the reference's list


def test_class_1_3B(base_model_type):
    return base_of(base_model_type) in ("t2v_1.3B", "vace_1.3B")


def test_vace(base_model_type):
    return base_of(base_model_type) in ("vace_14B", "vace_1.3B")


def test_wan_5B(base_model_type):
    return base_of(base_model_type) in ("ti2v_2_2",)


def test_i2v_2_2(base_model_type):
    return base_of(base_model_type) in ("i2v_2_2",)


class family_handler():
    @staticmethod
    def query_supported_types():
        return [k + SUFFIX for k in _ARCH]

    @staticmethod
    def query_family_maps():
        """(equivalence map, compatibility map) of wan_handler.py:81-107, restricted to the supported types."""
        eqv = {"t2v_1.3B" + SUFFIX: "t2v" + SUFFIX, "t2v_2_2" + SUFFIX: "t2v" + SUFFIX, "flf2v_720p" + SUFFIX: "i2v" + SUFFIX}
        comp = {"t2v" + SUFFIX: [t + SUFFIX for t in ("vace_14B", "vace_1.3B", "t2v_1.3B")], "i2v" + SUFFIX: ["flf2v_720p" + SUFFIX]}
        return eqv, comp

    @staticmethod
    def query_model_family():
        return "wan"

    @staticmethod
    def query_family_infos():
        return {"wan": (0, "Wan2.1"), "wan2_2": (1, "Wan2.2")}

    @staticmethod
    def query_model_def(base_model_type, model_def):
        """The properties wgp.py reads from a Wan model definition (wan_handler.py:216-1007), for the supported types: class
        flags, fps, frame grid, VAE block size, profile folders, the sampler / guidance / step-skipping capabilities."""
        b = base_of(base_model_type)
        i2v, t2v, vace, wan_5B = test_class_i2v(b), test_class_t2v(b), test_vace(b), test_wan_5B(b)
        multiple_submodels = "URLs2" in model_def
        group = "wan2_2" if (b in ("t2v_2_2", "ti2v_2_2") or test_i2v_2_2(b)) else "wan"
        if b == "t2v_2_2" or test_i2v_2_2(b):
            profiles_dir = "wan_2_2"
        elif i2v:
            profiles_dir = "wan_i2v"
        elif wan_5B:
            profiles_dir = "wan_2_2_5B"
        elif test_class_1_3B(b):
            profiles_dir = "wan_1.3B"
        else:
            profiles_dir = "wan"
        extra = {
            "riflex": True, "i2v_class": i2v, "t2v_class": t2v, "vace_class": vace, "wan_5B_class": wan_5B,
            "multitalk_class": False, "standin_class": False, "lynx_class": False, "alpha_class": False,
            "i2v_2_2": test_i2v_2_2(b), "color_correction": True,
            "vae_block_size": 32 if wan_5B else 16, "profiles_dir": [profiles_dir], "group": group, "fps": 24 if wan_5B else 16,
            "frames_minimum": 17 if vace else 5, "frames_steps": 4,
            # sliding windows (the reference claims them for t2v / i2v / 5B / VACE, wan_handler.py:369): wgp.py's window loop hands
            # generate() the previous window's last frames (`input_video` / the control video's prefix) and its last latent frames
            # (`overlapped_latents`), and takes `latent_slice` back.  Served where generate() has the reference's mechanics: the i2v
            # prefix video (re-noised injection + clean restore), the 5B model's timestep injection, VACE's pinned context overlap
            # with `overlap_noise` (any2video.py:775-782, :1060-1072, :1150-1152, :1517-1526, :1755-1761).  Plain t2v: not claimed.
            "sliding_window": bool(i2v or wan_5B or vace),
            "multiple_submodels": multiple_submodels, "guidance_max_phases": 3, "flow_shift": True, "cfg_zero": True, "cfg_star": True,
            "adaptive_projected_guidance": True,
            "tea_cache": not (b == "i2v_2_2" or wan_5B or multiple_submodels), "mag_cache": True,
            "sample_solvers": [("unipc", "unipc"), ("euler", "euler"), ("dpm++", "dpm++"), ("flowmatch causvid", "causvid"), ("lcm + ltx", "lcm")],
            "sub_parallel_windows": False,
            # normalized attention guidance (wan_handler.py:994) and the image prompt types (:956-978): Start / End image, Video to
            # continue, Last-frame options for the i2v models -- generate(image_start=, image_end=) / the prefix-video path
            # (not flf2v: its text branch carries the second image's CLIP tokens, which the NAG path of the forward driver does not serve)
            "NAG": (vace or t2v or i2v) and b != "flf2v_720p", "self_refiner": True, "perturbation": not vace,      # skip-layer guidance: not with VACE blocks
            # of the reference's "TVL" / "TSVL" / "SEVL": 'L' (continue the last video) is a sliding-window feature; 'V' (video to
            # continue) is served where generate() has the path -- the i2v prefix video and the 5B model's timestep injection
            "image_prompt_types_allowed": "T" if (vace or b in ("t2v", "t2v_2_2")) else ("TSV" if b == "ti2v_2_2" else ("SEV" if i2v else "")),
            # what the HIP path does not implement: offload, compile, in-app quantisation
            "compile": False, "no_quantization": True, "backend": "hip-gfx950",
        }
        # the text encoder's files (wan_handler.py:221-229, :295-299): wgp.py resolves `text_encoder_filename` -- what load_model is handed --
        # from these URLs by the user's quantisation setting and downloads into this folder (wgp.py:4051-4058); a model definition may
        # override either
        folder = "umt5-xxl"
        urls = ["https://huggingface.co/DeepBeepMeep/Wan2.1/resolve/main/umt5-xxl/" + f
                for f in ("models_t5_umt5-xxl-enc-bf16.safetensors", "models_t5_umt5-xxl-enc-quanto_int8.safetensors")]
        if model_def.get("text_encoder_URLs") is not None or model_def.get("text_encoder_folder") is not None:
            folder = model_def.get("text_encoder_folder") or folder
            urls = model_def.get("text_encoder_URLs") or urls
        extra["text_encoder_URLs"], extra["text_encoder_folder"] = urls, folder
        if multiple_submodels:
            extra["no_steps_skipping"] = True
        if i2v:
            extra["motion_amplitude"] = True
            extra["black_frame"] = True
        if i2v or wan_5B:       # test_oneframe_overlap (wan_handler.py:41-42, :996-997): windows of these types overlap by exactly one frame
            extra["sliding_window_defaults"] = {"overlap_min": 1, "overlap_max": 1, "overlap_step": 0, "overlap_default": 1}
        # video-to-video (the "G" letter generate() serves for every model, any2video.py:1004-1044): the UI's choice list exists for the
        # reference's t2v class (wan_handler.py:428-443) and, with the start image, for i2v_2_2 (:394-410)
        if t2v or b == "i2v_2_2":
            with_image = b == "i2v_2_2"
            extra["guide_custom_choices"] = {
                "choices": [("Use Text & Image Prompt Only" if with_image else "Use Text Prompt Only", ""),
                            ("Video to Video guided by Text Prompt & Image" if with_image else "Video to Video guided by Text Prompt", "GUV"),
                            ("Video to Video guided by Text/Image Prompt and Restricted to the Area of the Video Mask" if with_image else
                             "Video to Video guided by Text Prompt and Restricted to the Area of the Video Mask", "GVA")],
                "default": "", "show_label": False, "letters_filter": "GUVA", "label": "Video to Video"}
            extra["mask_preprocessing"] = {"selection": ["", "A"], "visible": False}
            if with_image:
                extra["i2v_v2v"] = True
                extra["extract_guide_from_window_start"] = True
        if vace:
            # the control-video inputs of the VACE types (wan_handler.py:814-839): without them wgp.py offers no control video, mask or
            # reference images for the type.  What they switch on is prepared by wgp.py (preprocessors, outpainting, padding, positioned
            # frames) and reaches generate() as input_frames(2) / input_masks(2) / input_ref_images / input_ref_masks, all served.
            # Not claimed: `v2i_switch_supported` (image outputs, image_mode = 1: refused by generate()).
            extra.update({
                "control_net_weight_name": "Vace", "control_net_weight_size": 2,
                "guide_preprocessing": {"selection": ["", "UV", "PV", "OV", "DV", "SV", "LV", "CV", "MV", "V", "PDV", "PSV", "PLV", "DSV", "DLV", "SLV"],
                                        "labels": {"V": "Use Vace raw format"}},
                "mask_preprocessing": {"selection": ["", "A", "NA", "XA", "XNA", "YA", "YNA", "WA", "WNA", "ZA", "ZNA"]},
                "image_ref_choices": {"choices": [("None", ""), ("People / Objects", "I"), ("Landscape followed by People / Objects (if any)", "KI"),
                                                  ("Positioned Frames followed by People / Objects (if any)", "FI")], "letters_filter": "KFI"},
                "background_removal_label": "Remove Backgrounds behind People / Objects, keep it for Landscape or Positioned Frames",
                "video_guide_outpainting": [0, 1], "pad_guide_video": True, "guide_inpaint_color": 127.5, "forced_guide_mask_inputs": True,
                "return_image_refs_tensor": True,
            })
        return extra

    @staticmethod
    def query_model_files(computeList, base_model_type, model_def=None):
        """wan_handler.query_model_files (:1016-1070) for the supported types: the files wgp.py fetches beside the model definition's
        own URLs -- the UMT5 tokenizer folder, the family's VAE, and for the Wan2.1 i2v class the CLIP tower's folder (same repositories,
        folders and names; a subset of the reference's list: the 2x-upscaling VAE is not served)."""
        b = base_of(base_model_type)
        files = [{"repoId": "DeepBeepMeep/Wan2.1", "sourceFolderList": ["umt5-xxl"],
                  "fileList": [["special_tokens_map.json", "spiece.model", "tokenizer.json", "tokenizer_config.json"]]}]
        if test_wan_5B(b):
            files.append({"repoId": "DeepBeepMeep/Wan2.2", "sourceFolderList": [""], "fileList": [["Wan2.2_VAE.safetensors"]]})
        elif _ARCH.get(b, {}).get("model_type") == "i2v":
            files.append({"repoId": "DeepBeepMeep/Wan2.1", "sourceFolderList": ["xlm-roberta-large", ""],
                          "fileList": [["models_clip_open-clip-xlm-roberta-large-vit-huge-14-bf16.safetensors", "sentencepiece.bpe.model",
                                        "special_tokens_map.json", "tokenizer.json", "tokenizer_config.json"], ["Wan2.1_VAE.safetensors"]]})
        else:
            files.append({"repoId": "DeepBeepMeep/Wan2.1", "sourceFolderList": [""], "fileList": [["Wan2.1_VAE.safetensors"]]})
        return files

    @staticmethod
    def get_rgb_factors(base_model_type):
        """wan_handler.get_rgb_factors (:1009-1014): the latent -> RGB preview factors are a table of the host application
        (`shared/RGB_factors.py`); outside it there is no preview (wgp.py:8349-8352 takes None for "no preview")."""
        try:
            from shared.RGB_factors import get_rgb_factors
        except ImportError:
            return None, None
        b = base_of(base_model_type)
        return get_rgb_factors("wan", "ti2v_2_2" if test_wan_5B(b) else b)

    @staticmethod
    def load_model(model_filename, model_type, base_model_type, model_def, quantizeTransformer=False, text_encoder_quantization=None,
                   dtype=torch.bfloat16, VAE_dtype=torch.float16, mixed_precision_transformer=False, save_quantized=False,
                   submodel_no_list=None, text_encoder_filename=None, VAE_upsampling=None, checkpoint_dir="ckpts", device="cuda",
                   state_dicts=None, vae_state_dict=None, text_encoder=None, clip=None, **kwargs):
        """wan_handler.load_model (:1116-1158) for the HIP backend.  `model_filename`: the checkpoint path(s) wgp.py resolved
        (one per expert for Wan2.2).  Returns (WanAny2VHIP, {"pipe": {...}}).  VAE_dtype: what wgp.py:4038 passes -- torch.float16
        unless the user set `vae_precision` "32" (the default HERE is wgp.py's, not the reference signature's torch.float32: the
        fp32 plan is an exact but minute-class option).
        Test hooks: `state_dicts` / `vae_state_dict` / `text_encoder` bypass the file reads."""
        if quantizeTransformer or save_quantized:
            raise NotImplementedError("on-the-fly quantisation is part of the reference's low-VRAM machinery; the HIP backend loads "
                                      "bf16 or scaled-fp8 checkpoints as they are")
        from .checkpoint import read_safetensors, read_wan_file
        from .model import WanModelHIP
        from .pipeline import WanAny2VHIP
        b = base_of(base_model_type)
        if b not in _ARCH:
            raise ValueError(f"model type {base_model_type!r} is not supported by the HIP Wan handler ({sorted(_ARCH)})")
        arch = _ARCH[b]
        files = [model_filename] if isinstance(model_filename, str) else list(model_filename or [])
        # wgp.py:4003-4027: [expert 1, (expert 2,) module files ...] with submodel numbers [1, (2,) ...]; a module file (e.g. the
        # VACE blocks kept apart from the t2v base) carries 0 = shared by both experts, 1 / 2 = that expert's
        # (any2video.py:208-222: fast_load_transformers_model(main file, modules=[...]) reads them into the same module)
        nos = list(submodel_no_list) if submodel_no_list else [1, 2][:len(files)]
        if len(nos) != len(files) and state_dicts is None:
            raise ValueError(f"load_model: {len(files)} files but submodel_no_list has {len(nos)} entries")
        n_main = 2 if (len(nos) >= 2 and nos[1] == 2) else 1
        if state_dicts is not None:
            sds = list(state_dicts)
        else:
            # (wgp.py picks the file by the user's quantisation setting, get_model_filename wgp.py:2927: the quanto int8 form is the default)
            sds = [read_wan_file(f, device) for f in files[:n_main]]
            for f, no in zip(files[n_main:], nos[n_main:]):
                extra = read_wan_file(f, device)
                for k, sd in enumerate(sds):
                    if no in (0, k + 1):
                        sd.update(extra)
        if not sds:
            raise ValueError("load_model: no checkpoint given")
        # mixed_precision_transformer (wgp.py:4039 server setting "mixed_precision" -> any2video.py:190 lock_layers_dtypes(torch.float32)):
        # the time MLP, the time projection and every norm3 are registered in fp32 and the library runs its fp32-stream plan (csrc/mixed_ops.hip;
        # tests/test_gpu_mixed.py against the reference's own forward under those locks).  Served for the t2v / i2v2_2 / ti2v2_2 block chain
        # and (round 6) the Wan2.1 i2v / flf2v CLIP branch; WanModelHIP refuses the combination with VACE blocks; TeaCache / MagCache run in the plan
        # (fp32 residual, round 5).
        models = [WanModelHIP(device=device, mixed_precision=bool(mixed_precision_transformer), **arch).load_state_dict(sd) for sd in sds[:2]]
        # any2video.py:137-163: VAE_URLs of the model definition (a path, or a list whose first entry the host's file locator resolves),
        # else the family's default file, resolved by the host application's locator (its checkpoint folders are configurable) or, outside
        # it, under `checkpoint_dir`
        wan_5B = test_wan_5B(b)
        if wan_5B:
            from .vae22 import Wan22VAEHIP as VAE
        else:
            from .vae import WanVAEHIP as VAE
        vae_url = (model_def or {}).get("VAE_URLs", None)
        if isinstance(vae_url, str):
            vae_path = vae_url
        elif isinstance(vae_url, list) and len(vae_url):
            vae_path = _locate(vae_url[0], checkpoint_dir)
        else:
            vae_path = _locate("Wan2.2_VAE.safetensors" if wan_5B else "Wan2.1_VAE.safetensors", checkpoint_dir)
        vae = None
        # VAE_dtype: wgp.py:4038 passes torch.float16 for `vae_precision` "16" (its default) and torch.float for "32".  Both VAEs serve
        # both plans (fp32: csrc/vae_f32.hip + the `_f32` pieces of vae22_ops.hip since round 6, a slow exact option); any other dtype runs
        # the fp16 plan
        vae_kw = {"dtype": torch.float32} if VAE_dtype == torch.float32 else {}
        if vae_state_dict is not None:
            vae = VAE(state_dict=vae_state_dict, device=device, **vae_kw)
        elif os.path.isfile(vae_path):
            vae = VAE(vae_pth=vae_path, device=device, **vae_kw)
        # under wgp.py (real checkpoint files, no test hooks) a missing VAE / text-encoder file must fail HERE, not as an opaque
        # error after a full denoise: generate() would return latents with x = None, or fail on `context`
        from_files = state_dicts is None
        if vae is None and from_files:
            raise FileNotFoundError(f"load_model: VAE checkpoint not found ({vae_path})")
        if text_encoder is None and from_files and not (text_encoder_filename and os.path.isfile(str(text_encoder_filename))):
            raise FileNotFoundError(f"load_model: text-encoder checkpoint not found ({text_encoder_filename!r})")
        if text_encoder is None and text_encoder_filename and os.path.isfile(str(text_encoder_filename)):
            # any2video.py:128-134: T5EncoderModel(text_len, checkpoint, tokenizer_path = <checkpoint's folder>)
            from .t5 import T5EncoderModelHIP
            from .tokenizers import HuggingfaceTokenizer
            tok = HuggingfaceTokenizer(name=os.path.dirname(str(text_encoder_filename)), seq_len=512, clean="whitespace")
            te_sd = read_safetensors(text_encoder_filename)
            if any(k.endswith("._data") for k in te_sd):            # models_t5_umt5-xxl-enc-quanto_int8.safetensors
                from .checkpoint import dequantize_quanto_
                dequantize_quanto_(te_sd, device=device)
            text_encoder = T5EncoderModelHIP(512, tok, state_dict=te_sd, device=device)
        pipe = WanAny2VHIP(models[0], models[1] if len(models) > 1 else None, vae=vae, text_encoder=text_encoder, device=device,
                           vae_stride=(4, 16, 16) if test_wan_5B(b) else (4, 8, 8))
        pipe.model_def, pipe.base_model_type = model_def, base_model_type
        # Wan2.1 i2v class (any2video.py:127-132, :945-954): the CLIP visual tower is the host application's own module -- this plugin runs
        # inside it -- built as the reference builds it and handed to its offload profile under the reference's name (wan_handler.py:1156-1157);
        # one ViT-H forward on 257 tokens per video, outside the denoise path.  Standalone (no host application): generate(clip_fea=...)
        modules = {}
        if arch["model_type"] == "i2v":
            pipe.flf = b == "flf2v_720p"
            pipe.clip = clip if clip is not None else (_host_clip(device) if from_files else None)
            if getattr(pipe.clip, "model", None) is not None and isinstance(pipe.clip.model, torch.nn.Module):
                modules["text_encoder_2"] = pipe.clip.model
        # wgp.py:4074-4076: a handler may return {"pipe": modules_for_mmgp, **kwargs}; the HIP models themselves are not offloaded
        return pipe, {"pipe": modules}

    @staticmethod
    def get_lora_dir(base_model_type, args, lora_root):
        """wan_handler.get_lora_dir (:150-168; wgp.py:2482-2490 asks the handler): the HIP types read LoRAs from the same folders
        as the built-in Wan types they stand for -- `wan_i2v` for the Wan2.1 i2v model, `wan_1.3B`, `wan_5B`, else `wan` -- with
        the same command-line overrides.  (`register_lora_cli_args` is left to the built-in handler: the flags exist once.)"""
        b = base_of(base_model_type)
        i2v = test_class_i2v(b) and not test_i2v_2_2(b)

        def folder(default, *flags):                       # first command-line override that is set, else the folder under lora_root
            return next((v for v in (getattr(args, f, None) for f in flags) if v), os.path.join(lora_root, default))
        if i2v:
            return folder("wan_i2v", "lora_dir_wan_i2v", "lora_dir_i2v")
        if "1.3B" in b:
            return folder("wan_1.3B", "lora_dir_wan_1_3b")
        if test_wan_5B(b):
            return folder("wan_5B", "lora_dir_wan_5b")
        return folder("wan", "lora_dir_wan", "lora_dir")

    @staticmethod
    def set_cache_parameters(cache_type, base_model_type, model_def, inputs, skip_steps_cache):
        """wan_handler.set_cache_parameters (:172-214; called from wgp.py:7079 when step skipping is switched on): hands the
        per-model calibration data to the cache object -- MagCache magnitude ratios (+ threshold 0, K 2) or the TeaCache rescale
        polynomial, picked by model class and, for the Wan2.1 i2v model, by resolution.  The tables are the reference's literals
        (wan2gp_amd/data/skip_cache_tables.json, extracted by tools/extract_cache_tables.py)."""
        import json
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "skip_cache_tables.json")) as f:
            tables = json.load(f)
        b = base_of(base_model_type)
        i2v = test_class_i2v(b)
        width, height = inputs["resolution"].split("x")
        pixels = int(width) * int(height)
        if cache_type == "mag":
            skip_steps_cache.update({"magcache_thresh": 0, "magcache_K": 2})
            if b == "t2v" and "URLs2" in model_def:
                key = "t2v_two_experts"
            elif b == "i2v_2_2":
                key = "i2v_2_2"
            elif test_wan_5B(b):
                both = inputs.get("image_start", None) is not None and inputs.get("video_source", None) is not None
                key = "ti2v_5B_with_start_image_and_source_video" if both else "ti2v_5B"
            elif test_class_1_3B(b):
                key = "t2v_1.3B"
            elif i2v:
                key = "i2v_720p" if pixels >= 1280 * 720 else "i2v_480p"
            else:
                key = "t2v_14B"
            skip_steps_cache.def_mag_ratios = list(tables["mag_ratios"][key])
        else:
            if i2v:
                key = "i2v_720p" if pixels >= 1280 * 720 else "i2v_480p"
            else:
                key = "t2v_1.3B" if test_class_1_3B(b) else "t2v_14B"
            skip_steps_cache.coefficients = list(tables["tea_coefficients"][key])

    @staticmethod
    def fix_settings(base_model_type, settings_version, model_def, ui_defaults):
        """Migration of saved settings of older versions (wan_handler.fix_settings, :1161-1250), the branches that apply to the
        supported types: empty solver name, guidance phases of two-expert models, three-phase LoRA multipliers, the one-frame
        sliding-window overlap of the i2v / 5B models, the start-image default, a self-refiner plan stored as a list."""
        b = base_of(base_model_type)
        if ui_defaults.get("sample_solver", "") == "":
            ui_defaults["sample_solver"] = "unipc"
        if settings_version < 2.24:
            if (model_def.get("multiple_submodels", False) or ui_defaults.get("switch_threshold", 0) > 0) and ui_defaults.get("guidance_phases", 0) < 2:
                ui_defaults["guidance_phases"] = 2
        if settings_version == 2.24 and ui_defaults.get("guidance_phases", 0) == 2:
            mult = model_def.get("loras_multipliers", "")
            if len(mult) > 1 and len(mult[0].split(";")) == 3:
                ui_defaults["guidance_phases"] = 3
        if settings_version < 2.31 and (test_class_i2v(b) or test_wan_5B(b)):          # test_oneframe_overlap
            ui_defaults["sliding_window_overlap"] = 1
        if settings_version < 2.32:
            if test_class_i2v(b) and len(ui_defaults.get("image_prompt_type", "")) == 0 and "S" in model_def.get("image_prompt_types_allowed", ""):
                ui_defaults["image_prompt_type"] = "S"
        if model_def.get("self_refiner", False) and settings_version < 2.47:
            ui_defaults["self_refiner_setting"] = 0
            ui_defaults["self_refiner_plan"] = ""
        if model_def.get("self_refiner", False) and settings_version < 2.48:
            ui_defaults["self_refiner_f_uncertainty"] = 0.1
            ui_defaults["self_refiner_certain_percentage"] = 0.999

    @staticmethod
    def update_default_settings(base_model_type, model_def, ui_defaults):
        """Defaults of wan_handler.update_default_settings (:1252-1455) for the supported types."""
        b = base_of(base_model_type)
        ui_defaults.update({"sample_solver": "unipc"})
        if test_class_i2v(base_model_type) and "S" in model_def.get("image_prompt_types_allowed", ""):
            ui_defaults["image_prompt_type"] = "S"
        if b == "vace_14B":                                                             # :1317-1320
            ui_defaults.update({"sliding_window_discard_last_frames": 0})
        elif b == "ti2v_2_2":                                                           # :1322-1325
            ui_defaults.update({"image_prompt_type": "T"})
        if test_wan_5B(b):                                                              # :1436-1439
            ui_defaults.update({"sliding_window_size": 121})
        if b == "i2v_2_2":                                                              # :1441-1442
            ui_defaults.update({"masking_strength": 0.1, "denoising_strength": 0.9})
        if test_class_i2v(b) or test_wan_5B(b):                                         # test_oneframe_overlap (:41-42, :1447-1449)
            ui_defaults["sliding_window_overlap"] = 1
            ui_defaults["sliding_window_color_correction_strength"] = 0
        if model_def.get("multiple_submodels", False):                                  # :1454-1455
            ui_defaults["guidance_phases"] = 2

    @staticmethod
    def validate_generative_settings(base_model_type, model_def, inputs):
        if inputs.get("sample_solver", "unipc") not in ("unipc", "", "euler", "dpm++", "causvid", "lcm"):
            return f"Unsupported sample solver {inputs.get('sample_solver')!r}"
        # what generate() would refuse in the middle of a run is refused here, where wgp.py shows the message (wgp.py:1056-1070)
        if (inputs.get("image_mode", 0) or 0) == 1:
            return "The HIP backend generates videos only (image outputs, image_mode 1, are not served)."
        return None
