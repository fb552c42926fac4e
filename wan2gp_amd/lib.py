"""ctypes binding of libwanhip.so (the C ABI in include/wanhip.h).

PyTorch is used only as the owner of device memory and streams: every call passes raw
``data_ptr()`` values and the current HIP stream.  There is NO fallback: if the shared
library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwanhip.so")

EPI_NONE, EPI_GELU_TANH, EPI_GATE_RES, EPI_TRANSPOSED = 0, 1, 2, 3
WAN_ABORTED = 100     # wan_dit_forward*: stopped by the interrupt poll (include/wanhip.h)

_lib = None


class WanHipError(RuntimeError):
    pass


class DitConfig(ctypes.Structure):
    _fields_ = [("dim", c_int), ("ffn_dim", c_int), ("num_heads", c_int), ("num_layers", c_int),
                ("in_dim", c_int), ("out_dim", c_int), ("text_dim", c_int), ("freq_dim", c_int),
                ("text_len", c_int), ("eps", c_float)]


POLL_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int)


class DitArgs(ctypes.Structure):
    """wan_dit_args (include/wanhip.h)."""
    _fields_ = [("S", c_int), ("x", POINTER(c_void_p)), ("t", c_float), ("context", POINTER(c_void_p)), ("y", c_void_p),
                ("cos", c_void_p), ("sin", c_void_p), ("outs", POINTER(c_void_p)), ("F", c_int), ("H", c_int), ("W", c_int),
                ("workspace", c_void_p), ("workspace_bytes", c_int64), ("sp", c_void_p), ("poll", c_void_p),
                ("poll_user", c_void_p), ("should_calc", POINTER(c_int)), ("residual", POINTER(c_void_p)),
                ("vace_context", c_void_p), ("vace_scale", c_float), ("t_frames", POINTER(c_float)), ("n_t_frames", c_int),
                ("n_vace", c_int), ("vace_contexts", POINTER(c_void_p)), ("vace_scales", POINTER(c_float)),
                ("nag_scale", c_float), ("nag_tau", c_float), ("nag_alpha", c_float), ("context_batches", POINTER(c_int)),
                ("perturbation_layers", POINTER(c_int)), ("n_perturbation_layers", c_int), ("x_id", c_int),
                ("context_key", ctypes.c_uint64)]
GATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p)
GATHER_WAIT_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_void_p)


SP_ALLGATHER, SP_ULYSSES = 0, 1


class SpInfo(ctypes.Structure):
    _fields_ = [("rank", c_int), ("world", c_int), ("tok0", c_int64), ("tok_local", c_int64),
                ("gather_begin", GATHER_FN), ("gather_wait", GATHER_WAIT_FN), ("user", c_void_p),
                ("mode", c_int), ("a2a_begin", GATHER_FN), ("a2a_wait", GATHER_WAIT_FN), ("a2a_chunks", c_int)]


# name -> (restype, argtypes); the single source of truth mirrored by tests/test_abi.py
SIGNATURES = {
    "wan_last_error": (c_char_p, []),
    "wan_version": (c_int, []),
    "wan_device_cus": (c_int, []),
    "wan_rmsnorm_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                 c_int64, c_int, c_float, c_void_p]),
    "wan_rmsnorm_rope_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                        c_int64, c_int, c_float, c_float, c_void_p]),
    "wan_rmsnorm_rope_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_float, c_float,
                                      c_int, c_int, c_int, c_void_p]),
    "wan_ln_modulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64,
                                c_int, c_float, c_void_p]),
    "wan_ln_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "wan_nag_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_void_p]),
    "wan_gated_residual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int,
                                   c_void_p]),
    "wan_gemm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                              c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "wan_fp8_quantize": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_fp8_quantize_pre": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "wan_ln_modulate_amax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int, c_float, c_void_p,
                                     c_int64, c_void_p]),
    "wan_ln_affine_amax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_int64, c_void_p]),
    "wan_gemm_fp8_amax": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p,
                                  c_void_p]),
    "wan_gemm_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_int,
                             c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "wan_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64,
                              c_int, c_void_p]),
    "wan_attention_seg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64,
                                  c_int, c_int, c_int64, c_int64, c_void_p]),
    "wan_attention_prescaled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64,
                                        c_int, c_int, c_int64, c_int64, c_void_p]),
    "wan_attention_bounded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64,
                                      c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "wan_attention_scratch_words": (c_int64, [c_int, c_int, c_int64, c_int]),
    "wan_attention_raw_words": (c_int64, [c_int, c_int64, c_int]),
    "wan_attention_sp_local": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "wan_attention_sp_remote": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int64,
                                        c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "wan_attention_qscale": (c_float, []),
    "wan_transpose_v": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p]),
    "wan_t5_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "wan_mul_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_vae22_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_vae22_to_video": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_vae22_avgdown_add": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_vae22_dupup_add": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_vae22_patchify_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_vae22_avgdown_add_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_vae22_dupup_add_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_dit_set_clip": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wan_dit_forward_ex": (c_int, [c_void_p, POINTER(DitArgs), c_void_p]),
    "wan_dit_forward_graph": (c_int, [c_void_p, POINTER(DitArgs), c_void_p, POINTER(c_int)]),
    "wan_dit_set_vace_layers": (c_int, [c_void_p, POINTER(c_int), c_int]),
    "wan_dit_set_vace_contexts": (c_int, [c_void_p, c_int]),
    "wan_axpy_bf16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "wan_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_sub_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_lora_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "wan_axpy_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "wan_add_f32_into_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_dequant_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "wan_patch_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_int, c_void_p]),
    "wan_head": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                         c_int, c_int, c_int, c_float, c_void_p]),
    "wan_unpatchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_unpatchify_n": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_head_n": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                           c_int, c_float, c_int, c_void_p]),
    "wan_sinusoid": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "wan_act_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "wan_gemv_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "wan_lincomb": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_float), c_int64, c_void_p]),
    "wan_cfg_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "wan_sp_unique_id": (c_int, [c_void_p]),
    "wan_sp_init": (c_int, [POINTER(c_void_p), c_int, c_int, c_void_p]),
    "wan_sp_destroy": (None, [c_void_p]),
    "wan_sp_gather_begin": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_sp_gather_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "wan_sp_a2a_begin": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_permute16": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "wan_permute16_ex": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "wan_vae_conv3d_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p] + [c_int] * 17 + [c_void_p]),
    "wan_vae_rmsnorm_silu_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "wan_gemm_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    "wan_vae_softmax_f32": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "wan_vae_pack_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "wan_vae_unpack_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "wan_sp_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_sched_create": (c_int, [POINTER(c_void_p), c_int, c_int]),
    "wan_sched_destroy": (None, [c_void_p]),
    "wan_sched_set_timesteps": (c_int, [c_void_p, c_int, c_double, POINTER(c_double), POINTER(c_float)]),
    "wan_sched_step": (c_int, [c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_vae_conv3d": (c_int, [c_void_p] * 7 + [c_int] * 17 + [c_void_p]),
    "wan_vae_conv3d_ex": (c_int, [c_void_p] * 8 + [c_int] * 17 + [c_void_p]),
    "wan_vae_debug_force_big": (c_int, [c_int]),
    "wan_vae_debug_no_halo": (c_int, [c_int]),
    "wan_attention_debug_no_persist": (c_int, [c_int]),
    "wan_attention_debug_split_tail": (c_int, [c_int]),
    "wan_gemm_debug_force16s": (c_int, [c_int]),
    "wan_gemm_debug_force_tile_rows": (c_int, [c_int]),
    "wan_mx_ln_modulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int, c_float, c_void_p]),
    "wan_mx_ln_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "wan_mx_gated_residual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_void_p]),
    "wan_mx_debug_generic_rows": (None, [c_int]),
    "wan_gemm_bf16_res32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                    c_int, c_int, c_int64, c_void_p]),
    "wan_mx_patch_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "wan_mx_sinusoid": (c_int, [c_float, c_void_p, c_int, c_void_p]),
    "wan_mx_linear_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_mx_head": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int64, c_int, c_void_p]),
    "wan_vae_create": (c_int, [POINTER(c_void_p)]),
    "wan_vae_destroy": (None, [c_void_p]),
    "wan_vae_set_conv": (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int]),
    "wan_vae_set_gamma": (c_int, [c_void_p, c_char_p, c_void_p, c_int]),
    "wan_vae_set_attention": (c_int, [c_void_p, c_char_p, c_void_p, c_void_p, c_int]),
    "wan_vae_workspace_bytes": (c_int64, [c_void_p, c_int, c_int, c_int, c_int]),
    "wan_vae_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_vae_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_vae_rmsnorm_silu": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "wan_gemm_f16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                             c_int, c_float, c_int, c_void_p]),
    "wan_vae_softmax": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "wan_vae_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "wan_vae_unpack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "wan_vae_to_video": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "wan_prof_enable": (c_int, [c_int]),
    "wan_prof_collect": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_int)]),
    "wan_prof_attention_declined": (c_int, [POINTER(c_int64), POINTER(c_int64)]),
    "wan_mfma_sustained_probe": (c_int, [c_int, POINTER(c_double), c_void_p]),
    "wan_debug_delay": (c_int, [c_double, c_void_p]),
    "wan_attention_count_declined": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "wan_dit_create": (c_int, [POINTER(DitConfig), POINTER(c_void_p)]),
    "wan_dit_destroy": (None, [c_void_p]),
    "wan_dit_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int64]),
    "wan_dit_workspace_bytes": (c_int64, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "wan_dit_forward": (c_int, [c_void_p, c_int, POINTER(c_void_p), c_float, POINTER(c_void_p), c_void_p, c_void_p,
                                c_void_p, POINTER(c_void_p), c_int, c_int, c_int, c_void_p, c_int64,
                                POINTER(SpInfo), POLL_FN, c_void_p, c_void_p]),
    "wan_dit_forward_skip": (c_int, [c_void_p, c_int, POINTER(c_void_p), c_float, POINTER(c_void_p), c_void_p, c_void_p,
                                c_void_p, POINTER(c_void_p), c_int, c_int, c_int, c_void_p, c_int64,
                                POINTER(SpInfo), POLL_FN, c_void_p, POINTER(c_int), POINTER(c_void_p), c_void_p]),
}


def load():
    """Loads libwanhip.so (no GPU needed to load); raises WanHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise WanHipError(
            f"{LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C wan2gp_amd/csrc). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    ab_build = os.path.basename(LIB_PATH) != "libwanhip.so"      # tools/*.py --lib <older build>: entry points added since are simply absent
    for name, (res, args) in SIGNATURES.items():
        if ab_build and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError -> a header symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().wan_last_error().decode("utf-8", "replace")
        raise WanHipError(f"{what}: rc={rc}: {msg}")


def ptr(t):
    """data_ptr of a torch tensor (or None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
