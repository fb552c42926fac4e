"""Checkpoint wire format -> resident HIP weights.

Replaces, for the Wan DiT, what the reference does through `mmgp.offload.fast_load_transformers_model`
(`models/wan/any2video.py:187-224`) with `preprocess_sd = WanModel.preprocess_sd_with_dtype`
(`models/wan/modules/model.py:913-941`) and, for Diffusers-named files, `rename_key_universal`
(`models/wan/convert_wan.py:19-76`):

  * `read_safetensors`       -- the on-disk format (`*_mbf16.safetensors`, `quanto_*_int8`): 8-byte little-endian header
                                length, JSON header {name: {dtype, shape, data_offsets}}, raw little-endian tensor bytes.
                                Tensors are zero-copy views of one mmap; they go to HBM once (288 GB: both 14B experts
                                stay resident, there is no offload tier to stage through).
  * `normalize_wan_keys`     -- preprocess_sd_with_dtype: prefix strip, `.block.` removal, fp8 norm weights upcast ...
  * `rename_diffusers_key`   -- convert_wan.py's Diffusers -> canonical Wan key map.
  * `dequantize_quanto_`     -- optimum-quanto qint8 weights (`<w>._data` int8 [N,K] x `<w>._scale` [N,1]) -> bf16 in HBM
                                (libwanhip `wan_dequant_i8`).  mmgp 3.7.12 (requirements.txt:2), which reads that format in the
                                reference, is not vendored there: the tensor naming follows optimum-quanto's published
                                `WeightQBytesTensor` serialisation; parity for this one step is unpinned.
  * `load_wan_checkpoint`    -- files -> `WanModelHIP.load_state_dict`.
"""
import json
import mmap
import re
import struct
from typing import Dict, Iterable, Optional

import torch

from .lib import WanHipError

_ST_DTYPES = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32, "F64": torch.float64, "I8": torch.int8,
              "U8": torch.uint8, "I16": torch.int16, "I32": torch.int32, "I64": torch.int64, "BOOL": torch.bool,
              "F8_E4M3": torch.float8_e4m3fn, "F8_E5M2": torch.float8_e5m2}


def read_safetensors(path, with_metadata=False):
    """{name: CPU tensor} (views of one read-only mmap) [+ the `__metadata__` dict]."""
    f = open(path, "rb")
    try:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    finally:
        f.close()
    if len(mm) < 8:
        raise WanHipError(f"{path}: not a safetensors file (shorter than its 8-byte header length)")
    (n,) = struct.unpack("<Q", mm[:8])
    if n > len(mm) - 8:
        raise WanHipError(f"{path}: header length {n} exceeds the file")
    header = json.loads(mm[8:8 + n].decode("utf-8"))
    meta = header.pop("__metadata__", {}) or {}
    base = 8 + n
    buf = memoryview(mm)
    out = {}
    for name, info in header.items():
        dt = _ST_DTYPES.get(info["dtype"])
        if dt is None:
            raise WanHipError(f"{path}: tensor {name!r} has unsupported dtype {info['dtype']}")
        b, e = info["data_offsets"]
        shape = tuple(info["shape"])
        numel = 1
        for s in shape:
            numel *= s
        if e - b != numel * torch.empty((), dtype=dt).element_size() or base + e > len(mm):
            raise WanHipError(f"{path}: tensor {name!r} byte range [{b},{e}) does not match {info['dtype']} {list(shape)}")
        if numel == 0:
            out[name] = torch.empty(shape, dtype=dt)
        else:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")            # read-only buffer: the tensors are never written on the host
                out[name] = torch.frombuffer(buf[base + b: base + e], dtype=dt).view(shape)
    return (out, meta) if with_metadata else out


# ---------------------------------------------------------------------------------------------------------------------
_FP8 = (torch.float8_e5m2, torch.float8_e4m3fn)
_NORM_ENDS = (".norm3.bias", ".norm3.weight", ".norm_q.bias", ".norm_q.weight", ".norm_k.bias", ".norm_k.weight")


def normalize_wan_keys(sd: Dict[str, torch.Tensor], dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """`WanModel.preprocess_sd_with_dtype(dtype, sd)` (model.py:913-941): strip `model.diffusion_model.`, drop
    `.attn2.norm_added_q.` and `vae.*`, upcast fp8 norm vectors, pose/mask patch-embedding renames, `blocks.N.block.` ->
    `blocks.N.`.  Order of the surviving keys is preserved."""
    out = {}
    for k, v in sd.items():
        if k.startswith("model.diffusion_model"):
            k = k[len("model.diffusion_model") + 1:]
        if ".attn2.norm_added_q." in k:
            continue
        if v is not None and getattr(v, "dtype", None) in _FP8 and k.endswith(_NORM_ENDS):
            v = v.to(dtype)
        if k.startswith("patch_embedding_pose."):
            k = "pose_patch_embedding." + k[len("patch_embedding_pose."):]
        if k.startswith("patch_embedding_mask."):
            k = "mask_patch_embedding." + k[len("patch_embedding_mask."):]
        if k.startswith("blocks."):
            parts = k.split(".")
            if len(parts) > 2 and parts[2] == "block":
                k = ".".join(parts[:2] + parts[3:])
        if not k.startswith("vae."):
            out[k] = v
    return out


_BLK = r"^blocks\.(\d+)\."
_ATTN_SUBS = [(re.compile(r"(\b[^.\s]*attn[^.\s]*\b.*?\.)" + a + r"\."), r"\1" + b + ".")
              for a, b in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), (r"to_out\.0", "o"))]
_BLOCK_SUBS = [(re.compile(_BLK + a), r"blocks.\1." + b) for a, b in (
    (r"ffn\.net\.0\.proj\.", "ffn.0."), (r"ffn\.net\.2\.", "ffn.2."),
    (r"cross_attn\.add_k_proj\.", "cross_attn.k_img."), (r"cross_attn\.add_v_proj\.", "cross_attn.v_img."),
    (r"cross_attn\.norm_added_k\.", "cross_attn.norm_k_img."), (r"scale_shift_table$", "modulation"), (r"norm2\b", "norm3"))]
_TOP_SUBS = [(re.compile("^" + a), b) for a, b in (
    (r"condition_embedder\.text_embedder\.linear_1\.", "text_embedding.0."),
    (r"condition_embedder\.text_embedder\.linear_2\.", "text_embedding.2."),
    (r"condition_embedder\.time_embedder\.linear_1\.", "time_embedding.0."),
    (r"condition_embedder\.time_embedder\.linear_2\.", "time_embedding.2."),
    (r"condition_embedder\.time_proj\.", "time_projection.1."),
    (r"condition_embedder\.image_embedder\.norm1\.", "img_emb.proj.0."),
    (r"condition_embedder\.image_embedder\.ff\.net\.0\.proj\.", "img_emb.proj.1."),
    (r"condition_embedder\.image_embedder\.ff\.net\.2\.", "img_emb.proj.3."),
    (r"condition_embedder\.image_embedder\.norm2\.", "img_emb.proj.4."),
    (r"proj_out\.", "head.head."))]


def rename_diffusers_key(k: str) -> str:
    """`rename_key_universal` (convert_wan.py:36-76): Diffusers `WanTransformer3DModel` names -> the canonical Wan names
    `WanModelHIP.load_state_dict` takes.  Canonical names pass through unchanged."""
    k = re.sub(_BLK + r"attn1\.", r"blocks.\1.self_attn.", k)
    k = re.sub(_BLK + r"attn2\.", r"blocks.\1.cross_attn.", k)
    for rx, rep in _ATTN_SUBS:
        k = rx.sub(rep, k)
    for rx, rep in _BLOCK_SUBS[:2]:
        k = rx.sub(rep, k)
    for rx, rep in _BLOCK_SUBS[2:]:
        k = rx.sub(rep, k)
    for rx, rep in _TOP_SUBS:
        k = rx.sub(rep, k)
    return "head.modulation" if k == "scale_shift_table" else k


def convert_diffusers_state_dict(sd, cast_dtype: Optional[str] = None):
    """`convert_state_dict_universal` (convert_wan.py:80-93)."""
    dtype = None
    if cast_dtype:
        cd = cast_dtype.lower().strip()
        table = {"float16": torch.float16, "fp16": torch.float16, "half": torch.float16, "bfloat16": torch.bfloat16,
                 "bf16": torch.bfloat16, "float32": torch.float32, "fp32": torch.float32}
        if cd not in table:
            raise ValueError(f"Unsupported cast_dtype: {cast_dtype}")
        dtype = table[cd]
    return {rename_diffusers_key(k): (t.to(dtype) if dtype is not None else t) for k, t in sd.items()}


# ---------------------------------------------------------------------------------------------------------------------
def dequantize_quanto_(sd: Dict[str, torch.Tensor], device="cuda") -> Dict[str, torch.Tensor]:
    """In place on the dict: every (`X._data` int8, `X._scale`) pair becomes `X` = bf16(data * scale) in HBM;
    the per-module `input_scale` / `output_scale` activations scales (unused for weight-only qint8) are dropped."""
    from . import ops
    for k in [k for k in sd if k.endswith("._data")]:
        name = k[: -len("._data")]
        data, scale = sd.pop(k), sd.pop(name + "._scale", None)
        if scale is None:
            raise WanHipError(f"quanto tensor {name!r}: `_data` without `_scale`")
        if data.dtype != torch.int8 or data.dim() != 2 or scale.numel() != data.shape[0]:
            raise WanHipError(f"quanto tensor {name!r}: expected int8 [N,K] data and a per-row scale, got {data.dtype} "
                              f"{list(data.shape)} / {list(scale.shape)}")
        sd[name] = ops.dequant_i8(data.to(device).contiguous(), scale.to(device=device, dtype=torch.float32).reshape(-1).contiguous())
    for k in [k for k in sd if k.endswith((".input_scale", ".output_scale"))]:
        del sd[k]
    return sd


def read_wan_file(path, device="cuda", dtype=torch.bfloat16, diffusers_names: Optional[bool] = None) -> Dict[str, torch.Tensor]:
    """One checkpoint file -> a state dict `WanModelHIP.load_state_dict` takes: Diffusers names converted (detected from
    `condition_embedder.` / `attn1.` keys unless told), the reference's key normalisation, quanto int8 pairs dequantised to bf16."""
    sd = read_safetensors(path)
    if diffusers_names or (diffusers_names is None and any(".attn1." in k or k.startswith("condition_embedder.") for k in sd)):
        sd = convert_diffusers_state_dict(sd)
    sd = normalize_wan_keys(sd, dtype)
    if any(k.endswith("._data") for k in sd):
        dequantize_quanto_(sd, device=device)
    return sd


def load_wan_checkpoint(model, paths: Iterable[str], dtype=torch.bfloat16, diffusers_names: Optional[bool] = None):
    """Files -> `model.load_state_dict` (a `WanModelHIP`): what `offload.fast_load_transformers_model(files,
    preprocess_sd=...)` amounts to for a resident model (any2video.py:187-224).  `diffusers_names=None` detects the
    Diffusers naming from the presence of `condition_embedder.` / `attn1.` keys."""
    if isinstance(paths, (str, bytes)):
        paths = [paths]
    for p in paths:
        model.load_state_dict(read_wan_file(p, model.device, dtype, diffusers_names))
    return model
