"""WanModelHIP -- drop-in for the reference `WanModel` on the t2v / i2v2_2 path.

Mirrors models/wan/modules/model.py:891-2150: same constructor keywords for the plain
DiT, same `forward(x, t, context, ..., y=, freqs=, pipeline=, callback=, x_id=, ...)`
list-in / list-out contract (x list is *consumed*, model.py:1558-1559; outputs are fp32
[B,16,F,H,W], :2093-2097; returns [None]*n when interrupted, :1997-1998; calls
`callback(-1, None, False, True)` between blocks, :1995-1996).  All arithmetic runs in
libwanhip (resident weights, no mmgp offload); variant conditioning kwargs that are not on
the hot path raise NotImplementedError unless a `reference_module` is attached to delegate to.
"""
import ctypes
from ctypes import c_void_p
from typing import Dict, List, Optional

import torch

from . import lib as _L
from .lib import DitConfig, POLL_FN, SpInfo, check, ptr, stream_ptr
from .rope import get_rotary_pos_embed

# checkpoint tensors the reference locks to fp32 (lock_layers_dtypes, model.py:1330-1371)
FP32_PREFIXES = ("patch_embedding.", "head.")
# `mixed_precision_transformer` (wgp.py:4039 -> any2video.py:190 -> lock_layers_dtypes(torch.float32), model.py:1338-1346): the time MLP, the
# time projection and every block's norm3 are kept in fp32 as well; the library then runs its mixed-precision plan (the residual stream, e / e0,
# every modulate and gated residual in fp32 between bf16 Linears) -- chosen, like in the reference, by the dtype of time_projection.1.weight
MIXED_FP32_PREFIXES = ("time_embedding.", "time_projection.")


def dequantize_scaled_fp8(weight, scale=None, dtype=torch.bfloat16):
    """ScaledFP8WeightTensor._linear_fallback's weight (shared/qtypes/scaled_fp8.py:318-335): `weights.to(dtype)`, then IN PLACE
    `*= scale.to(dtype)` reshaped to [N, 1, ...] when it has one entry per output row (_reshape_scale, :135-142) -- every product rounded to
    `dtype` once.  scale None: ScaledFP8WeightTensor.create's default of 1 (:229-230)."""
    out = weight.to(dtype)
    if scale is None:
        return out
    sc = scale.to(device=out.device, dtype=dtype)
    if sc.numel() == 1:
        return out.mul_(sc.reshape(()))
    if sc.numel() != out.shape[0]:
        raise ValueError(f"scale of {sc.numel()} entries for a weight of {out.shape[0]} rows")
    return out.mul_(sc.reshape(out.shape[0], *([1] * (out.dim() - 1))))


def _wants_fp32(key: str, mixed: bool) -> bool:
    return key.startswith(FP32_PREFIXES) or (mixed and (key.startswith(MIXED_FP32_PREFIXES) or (key.startswith("blocks.") and ".norm3." in key)))

# non-None defaults of the variant keywords of WanModel.forward (model.py:1485-1543); a call that
# leaves them at these values is the plain t2v / i2v2_2 path
_VARIANT_DEFAULTS = {"vace_context_scale": [1.0], "causal_block_size": 1, "causal_attention": False,
                     "ref_images_count": 0, "lynx_ip_scale": 0, "lynx_ref_scale": 0, "lynx_feature_extractor": False,
                     "kiwi_ref_pad_first": False, "animate2_log_scale": 0.0, "animate2_kv_cache": "Disabled"}


def _is_default(k, v):
    if v is None:
        return True
    if torch.is_tensor(v):
        return False
    return k in _VARIANT_DEFAULTS and v == _VARIANT_DEFAULTS[k]


class WanModelHIP:
    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, eps=1e-6, device="cuda",
                 vace_layers=None, vace_in_dim=None, mixed_precision=False, **unused):
        if tuple(patch_size) != (1, 2, 2):
            raise NotImplementedError("only patch_size (1,2,2) (all Wan 2.1/2.2 14B/1.3B models)")
        if model_type not in ("t2v", "i2v2_2", "ti2v2_2", "i2v"):
            raise NotImplementedError(f"model_type {model_type!r}: t2v / i2v2_2 / ti2v2_2 (t2v_cross_attn) and i2v (Wan2.1 "
                                      "i2v_cross_attn with CLIP tokens) are implemented (model.py:1149)")
        self.model_type, self.dim, self.ffn_dim, self.num_heads, self.num_layers = model_type, dim, ffn_dim, num_heads, num_layers
        self.in_dim, self.out_dim, self.text_dim, self.freq_dim, self.text_len, self.eps = in_dim, out_dim, text_dim, freq_dim, text_len, eps
        self.patch_size = tuple(patch_size)
        self.device = torch.device(device)
        self.mixed_precision = bool(mixed_precision)
        if self.mixed_precision and vace_layers is not None:
            raise NotImplementedError("mixed_precision: the fp32-stream plan serves the t2v / i2v2_2 / ti2v2_2 / i2v (CLIP) block chains; VACE "
                                      "context blocks run in the bf16 plan only")
        self.cache = None
        # the forward as a replayed launch list (wan_dit_forward_graph, csrc/dit.hip): "auto" = when the joint pass holds at most
        # graph_max_tokens tokens (launch-bound shapes: BASELINE configs[0] is 6,400; 480p x 81 frames is 65,520), "on" / "off";
        # last_graph_how = what the last forward did (0 eager, 1 eager first sight, 2 captured, 3 replayed)
        # Default "off" (round 6, advisor): the replay measured a null on the one launch-bound BASELINE shape twice (configs[0]: 34.4 ms replayed /
        # 34.2 eager in round 5, 25.93 / 25.91 in run 14 -- the step is GPU-bound) while every replayed forward pays a staging copy, an output
        # clone and, per key, a capture with scratch rings of its own.  "auto" / "on" remain for hosts whose CPU is slower than the bench box's.
        self.graph = "off"
        self.graph_max_tokens = 16384
        self.last_graph_how = 0
        # the text cache (wan_dit_args.context_key): cross-attention K / V^T and the text embedding of an unchanged prompt are kept across
        # forwards (bit-identical; 1.7 GB per 14B expert).  False = recompute them in every forward like the reference does.
        self.text_cache = True
        self._tc_keys, self._tc_counter = [], 0
        self._graph_stage = {}
        self._rope_cache = {}
        self.reference_module = None      # optional: the reference nn.Module to delegate variant calls to
        self.sp = None                    # optional sequence-parallel group (wan2gp_amd.sp.SequenceParallel)
        # normalized attention guidance (NAG_scale, NAG_tau, NAG_alpha), set by generate() (any2video.py:607); None: read
        # mmgp.offload.shared_state["_nag_*"] when the reference's own generate() drives this model
        self.nag = None
        self._weights: Dict[str, torch.Tensor] = {}
        self._ws = None
        cfg = DitConfig(dim, ffn_dim, num_heads, num_layers, in_dim, out_dim, text_dim, freq_dim, text_len, eps)
        h = c_void_p()
        check(_L.load().wan_dit_create(ctypes.byref(cfg), ctypes.byref(h)), "wan_dit_create")
        self._ctx = h
        # VACE (model.py:1178-1206): context blocks attached to the main blocks `vace_layers`
        self.vace_layers = None if vace_layers is None else [int(v) for v in vace_layers]
        self.vace_in_dim = (in_dim if vace_in_dim is None else vace_in_dim) if vace_layers is not None else None
        self._vace_max_ctx = 1                # hint-stream sets the C workspace is sized for (wan_dit_set_vace_contexts)
        if self.vace_layers is not None:
            arr = (ctypes.c_int * len(self.vace_layers))(*self.vace_layers)
            check(_L.load().wan_dit_set_vace_layers(self._ctx, arr, len(self.vace_layers)), "wan_dit_set_vace_layers")

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                _L.load().wan_dit_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Checkpoint keys as in models/wan/convert_wan.py:19-76 (an optional
        'model.diffusion_model.' prefix is stripped, model.py:913-941).  Tensors are moved to
        HBM once and stay resident: bf16 everywhere except patch_embedding/head (fp32)."""
        lib = _L.load()
        pending_e5m2 = {}
        for k, v in sd.items():
            if k.startswith("model.diffusion_model."):
                k = k[len("model.diffusion_model."):]
            if k.endswith("modulation.weight"):           # post-init form (model.py:1291-1303)
                k = k[: -len(".weight")]
            # scaled-fp8 checkpoints (QLinearScaledFP8._load_from_state_dict, shared/qtypes/scaled_fp8.py:563-637): fp8 weight +
            # fp32 `<name>.scale_weight` (alias `.weight_scale`); input / output scales are not used at inference
            if k.endswith((".input_scale", ".output_scale", ".comfy_quant")) or k == "scaled_fp8":
                continue
            if k.endswith((".scale_weight", ".weight_scale")):
                k = k[: k.rindex(".")] + ".scale_weight"
                t = v.detach().to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
                self._weights[k] = t
                check(lib.wan_dit_set_weight(self._ctx, k.encode(), ptr(t), 1, t.numel()), f"wan_dit_set_weight({k})")
                continue
            if v.dtype == torch.float8_e4m3fn:
                t = v.detach().to(device=self.device).contiguous()
                self._weights[k] = t
                check(lib.wan_dit_set_weight(self._ctx, k.encode(), ptr(t), 2, t.numel()), f"wan_dit_set_weight({k})")
                continue
            if v.dtype == torch.float8_e5m2:
                # scaled_float8_e5m2 (shared/qtypes/scaled_fp8.py:17,34-49): torch._scaled_mm takes no e5m2 x e5m2 product, so the reference's
                # probe (:197-221) leaves _FP8_MM_SUPPORT[e5m2] False and every such Linear runs _linear_fallback (:318-335) -- weights.to(bf16)
                # *= scale.to(bf16), then a bf16 matmul.  The weights never change: that product is formed once, below, when every key has
                # been seen (the scale may follow its weight), and the Linear is a bf16 Linear from then on.
                pending_e5m2[k] = v
                continue
            want = torch.float32 if _wants_fp32(k, self.mixed_precision) else torch.bfloat16
            if k.startswith("vace_patch_embedding."):
                # a bf16 Conv3d in the reference (lock_layers_dtypes, model.py:1351-1355); the fp32 patch-embed kernel gets
                # fp32 copies of the bf16 values: identical products, fp32 accumulation, one bf16 rounding of the result
                t = v.detach().to(device=self.device, dtype=torch.bfloat16).to(torch.float32).contiguous()
                self._weights[k] = t
                check(lib.wan_dit_set_weight(self._ctx, k.encode(), ptr(t), 1, t.numel()), f"wan_dit_set_weight({k})")
                continue
            t = v.detach().to(device=self.device, dtype=want).contiguous()
            self._weights[k] = t
            check(lib.wan_dit_set_weight(self._ctx, k.encode(), ptr(t), 1 if want == torch.float32 else 0, t.numel()),
                  f"wan_dit_set_weight({k})")
        for k, v in pending_e5m2.items():
            want = torch.float32 if _wants_fp32(k, self.mixed_precision) else torch.bfloat16
            t = dequantize_scaled_fp8(v.detach().to(device=self.device), self._weights.get(k[: k.rindex(".")] + ".scale_weight") if k.endswith(".weight") else None)
            t = t.to(want).contiguous()
            self._weights[k] = t
            check(lib.wan_dit_set_weight(self._ctx, k.encode(), ptr(t), 1 if want == torch.float32 else 0, t.numel()), f"wan_dit_set_weight({k})")
        return self

    # ---- step-skipping caches (model.py:1373-1482) --------------------------------------------------------
    def time_embedding(self, tval):
        """e = time_embedding(sinusoidal_embedding_1d(freq_dim, t)) as bf16 [1, dim] (model.py:1815-1817): the same three
        kernels wan_dit_forward runs."""
        from . import ops
        w = self._weights
        if self.mixed_precision:      # the fp32 lock covers the time MLP (model.py:1338-1346): fp32 e, as the reference's TeaCache sees it
            from . import mixed_ops as mx
            h = mx.linear_f32(mx.sinusoid(float(tval), self.freq_dim, device=self.device), w["time_embedding.0.weight"], w["time_embedding.0.bias"])
            return mx.linear_f32(h, w["time_embedding.2.weight"], w["time_embedding.2.bias"], silu_input=True)
        s = ops.sinusoid(torch.tensor([float(tval)], dtype=torch.float32, device=self.device), self.freq_dim)
        h = ops.silu(ops.gemv(s, w["time_embedding.0.weight"], w["time_embedding.0.bias"]))
        return ops.gemv(h, w["time_embedding.2.weight"], w["time_embedding.2.bias"])

    def _tea_e(self, tval, t_frames, tflat):
        """TeaCache's e (model.py:1812-1817, :1954): one row per timestep the call carries."""
        return self.time_embedding(tval) if t_frames is None else torch.cat([self.time_embedding(float(v)) for v in tflat], 0)

    def compute_teacache_threshold(self, start_step, timesteps=None, speed_factor=0):
        from . import skipcache
        return skipcache.compute_teacache_threshold(self.cache, start_step, [self.time_embedding(float(t)) for t in timesteps], speed_factor)

    def compute_magcache_threshold(self, start_step, timesteps=None, speed_factor=0):
        from . import skipcache
        return skipcache.compute_magcache_threshold(self.cache, start_step, timesteps, speed_factor)

    def _nag_params(self):
        """(scale, tau, alpha) of text_cross_attention's NAG branch (model.py:249, :263-264) or None when it is off."""
        nag = self.nag
        if nag is None:
            import sys
            off = sys.modules.get("mmgp.offload") or getattr(sys.modules.get("mmgp"), "offload", None)
            st = getattr(off, "shared_state", None)
            if st and st.get("_nag_scale", 0) > 1:
                nag = (st["_nag_scale"], st["_nag_tau"], st["_nag_alpha"])
        return None if nag is None or nag[0] <= 1 else tuple(float(v) for v in nag)

    def apply_post_init_changes(self):  # reference API; nothing to adapt here
        return self

    def eval(self):
        return self

    def debug_token_stream(self, S, L):
        """View of the bf16 token streams [S, L_local, dim] inside the forward workspace (first region carved by
        wan_dit_workspace_bytes).  Valid between blocks (inside `callback`) after a torch.cuda.synchronize(): the
        per-layer error-growth tables of tests/test_gpu_baseline_configs.py read it."""
        n = S * L * self.dim
        if self.mixed_precision:                       # the fp32 residual stream of the mixed-precision plan
            return self._ws[: n * 4].view(torch.float32).view(S, L, self.dim)
        return self._ws[: n * 2].view(torch.bfloat16).view(S, L, self.dim)

    # ---- forward ---------------------------------------------------------------------------------
    def _workspace(self, S, F, H, W, shards):
        need = _L.load().wan_dit_workspace_bytes(self._ctx, S, F, H, W, shards)
        if need < 0:
            raise _L.WanHipError(f"wan_dit_workspace_bytes failed for S={S} latent={F}x{H}x{W} shards={shards}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, x, t, context, y=None, freqs=None, pipeline=None, current_step_no=0, real_step_no=0, x_id=0,
                max_steps=0, callback=None, clip_fea=None, vace_context=None, vace_context_scale=None, perturbation_layers=None,
                **variant_kwargs):
        active = {k: v for k, v in variant_kwargs.items() if not _is_default(k, v)}
        vace_ts, vace_scales = None, None
        if vace_context is not None:
            if self.vace_layers is None:
                active["vace_context"] = vace_context
            else:                                           # one hint stream set per context, each with its own scale (model.py:1905-1912)
                vace_scales = [float(v) for v in ([1.0] if vace_context_scale is None else vace_context_scale)]
                if len(vace_scales) != len(vace_context) or not 1 <= len(vace_context) <= 8:
                    raise _L.WanHipError(f"{len(vace_context)} VACE contexts with {len(vace_scales)} scales (1..8 contexts, one scale each)")
                vace_ts = [u.to(device=self.device, dtype=torch.bfloat16).to(torch.float32).contiguous() for u in vace_context]   # u.to(weight.dtype)
                if len(vace_ts) > self._vace_max_ctx:       # the workspace holds one hint-stream set per context
                    check(_L.load().wan_dit_set_vace_contexts(self._ctx, len(vace_ts)), "wan_dit_set_vace_contexts")
                    self._vace_max_ctx = len(vace_ts)
        if self.model_type == "i2v":
            if clip_fea is None or y is None:
                raise _L.WanHipError("model_type 'i2v' needs clip_fea [1,257,1280] and y (model.py:1547)")
            cf = clip_fea.to(device=self.device, dtype=torch.bfloat16).contiguous()
            # flf2v_720p (a checkpoint whose img_emb carries emb_pos, model.py:878-887): the CLIP features of the start AND the end
            # image, [2,257,1280] (any2video.py:949-950)
            n_img = 2 if "img_emb.emb_pos" in self._weights else 1
            if cf.numel() != n_img * 257 * 1280:
                raise _L.WanHipError(f"clip_fea must be [{n_img},257,1280] for this checkpoint, got {list(clip_fea.shape)}")
            # img_emb(clip_fea): three small GEMMs + two LayerNorms on 257 tokens -- redone per call like the reference does
            # (a pointer-keyed cache would be fooled by the allocator handing the same address to a different tensor)
            check(_L.load().wan_dit_set_clip(self._ctx, ptr(cf), stream_ptr()), "wan_dit_set_clip")
        elif clip_fea is not None:
            active["clip_fea"] = clip_fea
        if active:
            if self.reference_module is not None:
                extra = {k: v for k, v in (("clip_fea", clip_fea), ("vace_context", vace_context),
                                           ("vace_context_scale", vace_context_scale)) if v is not None}
                return self.reference_module(x, t, context, y=y, freqs=freqs, pipeline=pipeline,
                                             current_step_no=current_step_no, real_step_no=real_step_no, x_id=x_id,
                                             max_steps=max_steps, callback=callback, **extra, **variant_kwargs)
            raise NotImplementedError(f"WanModelHIP.forward: variant arguments {sorted(active)} are outside what this model "
                                      "instance implements (t2v / i2v2_2 / ti2v2_2, i2v with clip_fea, VACE with vace_layers); "
                                      "attach `reference_module` to delegate")
        x_list = list(x)
        x.clear()                                           # model.py:1558-1559
        S = len(x_list)
        if any(xx.shape[0] != 1 for xx in x_list):
            raise NotImplementedError("each stream must have batch 1 (the reference's joint CFG pass)")
        _, C, F, H, W = x_list[0].shape
        if C != self.out_dim:
            raise _L.WanHipError(f"latent streams must have {self.out_dim} channels, got {C}")
        dev = self.device
        xs = [xx.to(device=dev, dtype=torch.float32).contiguous() for xx in x_list]
        ctxs = [c.to(device=dev, dtype=torch.bfloat16).contiguous() for c in context]
        if any(c.shape[-2] != self.text_len or c.shape[0] not in (1, 2) for c in ctxs):
            raise _L.WanHipError(f"context must be [1,{self.text_len},{self.text_dim}] per stream ([2,...] = positive ; negative "
                                 "prompt under normalized attention guidance)")
        # NAG (any2video.py:607-608; text_cross_attention model.py:260): a batch-2 context is (positive ; negative) prompt
        ctx_batches = [int(c.shape[0]) for c in ctxs]
        nag = self._nag_params() if max(ctx_batches) == 2 else None
        if max(ctx_batches) == 2 and nag is None:
            raise _L.WanHipError("a context of batch 2 needs normalized attention guidance (NAG_scale > 1): set model.nag = "
                                 "(scale, tau, alpha) as generate() does")
        # t: one timestep, or one per latent frame (model.py:1812-1818; ti2v image conditioning any2video.py:1496-1499,
        # diffusion forcing with a [1, F] tensor)
        tflat = t.detach().flatten().to(torch.float32).cpu()
        if tflat.numel() not in (1, F):
            raise _L.WanHipError(f"t must hold 1 or F={F} timesteps, got {tflat.numel()}")
        tval = float(tflat[0].item())
        t_frames = (ctypes.c_float * F)(*[float(v) for v in tflat]) if tflat.numel() == F and F > 1 else None
        yy = None if y is None else y.to(device=dev, dtype=torch.float32).contiguous()
        if freqs is None:                                   # (kept per shape: a replayed launch list needs the tables where they were)
            if (F, H, W) not in self._rope_cache:
                if len(self._rope_cache) >= 4:
                    self._rope_cache.clear()
                self._rope_cache[(F, H, W)] = tuple(f.to(device=dev, dtype=torch.float32).contiguous() for f in get_rotary_pos_embed((F, H, W)))
            freqs = self._rope_cache[(F, H, W)]
        cos, sin = (f.to(device=dev, dtype=torch.float32).contiguous() for f in freqs)

        sp = self.sp if (self.sp is not None and self.sp.world > 1) else None   # a group of one shards nothing: the plain forward
        shards = 1 if sp is None else sp.world
        ws = self._workspace(S, F, H, W, shards)
        L = F * (H // 2) * (W // 2)
        if sp is None:
            outs = [torch.empty(1, self.out_dim, F, H, W, dtype=torch.float32, device=dev) for _ in range(S)]
            sp_struct = None
        else:
            outs = [torch.empty(1, L // shards, 4 * self.out_dim, dtype=torch.float32, device=dev) for _ in range(S)]
            sp.bind_workspace(ws)
            sp_struct = ctypes.byref(sp.make_info(L, heads=self.num_heads))

        def _poll(user, block_idx):
            try:
                if callback is not None:
                    callback(-1, None, False, True)         # model.py:1995-1996
                return 1 if (pipeline is not None and getattr(pipeline, "_interrupt", False)) else 0
            except Exception:                                # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1

        poll = POLL_FN(_poll)
        XP = (c_void_p * S)(*[a.data_ptr() for a in xs])
        CP = (c_void_p * S)(*[a.data_ptr() for a in ctxs])
        OP = (c_void_p * S)(*[a.data_ptr() for a in outs])
        cache = self.cache
        FL = RP = None
        if cache is not None:
            # TeaCache / MagCache (model.py:1914-2064): host decision, residual bookkeeping inside the forward
            from . import skipcache
            # CFG parallelism (sp.CfgParallel): the conditional stream runs in another process, but what the reference's x_id-1 pass reads
            # from the x_id-0 pass of the same step -- TeaCache's should_calc, MagCache's one_for_all verdict (model.py:1921-1923, :1945-1946)
            # -- depends on the timestep, the weights and the cache's own counters only, never on the latents: the unconditional rank
            # makes the conditional stream's decision itself first (same state, same arithmetic, nothing exchanged)
            if getattr(self, "cfg_parallel_stream", None) == 1 and x_id == 1 and S == 1:
                skipcache.decide(cache, 1, 0, real_step_no, self._tea_e(tval, t_frames, tflat) if cache.cache_type == "tea" else None)
            e = None
            if cache.cache_type == "tea" and x_id == 0:
                # TeaCache decides on e = time_embedding(sinusoidal(t.flatten())) (model.py:1812-1817, :1954): one row per timestep, i.e. F rows
                # under per-frame timesteps (the relative L1 is a mean over whatever e holds).  The reference's handler switches TeaCache
                # off for the one model that uses per-frame timesteps (the 5B ti2v class; MagCache stays on) -- the forward serves it anyway
                e = self._tea_e(tval, t_frames, tflat)
            flags = skipcache.decide(cache, S, x_id, real_step_no, e)
            if getattr(cache, "previous_residual", None) is None:
                cache.previous_residual = [None] * S
            slots = list(range(S)) if S > 1 else [x_id]
            n_res = (L // shards) * self.dim
            rdt = torch.float32 if self.mixed_precision else torch.bfloat16     # the residual has the stream's dtype (model.py:2044-2062)
            bufs = []
            for sl, calc in zip(slots, flags):
                while len(cache.previous_residual) <= sl:
                    cache.previous_residual.append(None)
                r = cache.previous_residual[sl]
                if r is None or r.numel() != n_res or not r.is_cuda or r.dtype != rdt:
                    if not calc:
                        raise _L.WanHipError(f"step-skipping cache: stream {sl} is skipped at step {real_step_no} without a stored residual")
                    r = cache.previous_residual[sl] = torch.empty(n_res, dtype=rdt, device=dev)
                bufs.append(r)
            FL = (ctypes.c_int * S)(*[1 if f else 0 for f in flags])
            RP = (c_void_p * S)(*[r.data_ptr() for r in bufs])
        # skip-layer guidance (any2video.py:1502; model.py:2025-2028): blocks that run for the first stream of the x_id-0 call only
        slg = [int(v) for v in perturbation_layers] if perturbation_layers is not None else []
        if any(v < 0 or v >= self.num_layers for v in slg):
            raise _L.WanHipError(f"perturbation_layers {slg} outside [0, {self.num_layers})")
        use_graph = (self.graph == "on" or (self.graph == "auto" and S * L <= self.graph_max_tokens)) and sp is None and cache is None \
            and t_frames is None and not self.mixed_precision and vace_ts is None and callback is None   # (a per-block callback keeps its contract: eager)
        self.last_graph_how = 0
        if use_graph:
            # stable addresses for what changes from step to step (the scheduler hands new latent tensors, the outputs are fresh
            # allocations): staging copies of a few hundred KB; contexts / y / rope tables are the caller's own long-lived tensors
            key = (S, C, F, H, W)
            if key not in self._graph_stage:
                if len(self._graph_stage) >= 4:
                    self._graph_stage.clear()
                self._graph_stage[key] = ([torch.empty(1, C, F, H, W, dtype=torch.float32, device=dev) for _ in range(S)],
                                          [torch.empty(1, self.out_dim, F, H, W, dtype=torch.float32, device=dev) for _ in range(S)])
            sx, so = self._graph_stage[key]
            for a_, b_ in zip(sx, xs):
                a_.copy_(b_)
            XP = (c_void_p * S)(*[a_.data_ptr() for a_ in sx])
            OP = (c_void_p * S)(*[a_.data_ptr() for a_ in so])
        # The text cache (wan_dit_args.context_key; csrc/dit.hip TextCache): the text embedding and every block's cross-attention K / V^T
        # depend on the context tensors and the weights only.  The key names the contents of `context`: it changes when another tensor
        # object arrives or one of them was written in place (torch's version counter).  The tensors that produced the cached K / V^T are
        # kept alive here, so the allocator cannot hand their storage to a different prompt (what a pointer-keyed cache would fall for);
        # a context that had to be converted (dtype / device / layout) is a new object in every call and simply never hits.
        ctx_key = 0
        if self.text_cache:
            vers = tuple(int(c._version) for c in ctxs)
            hit = [i for i, (src, v, _) in enumerate(self._tc_keys) if len(src) == len(ctxs) and all(a is b for a, b in zip(src, ctxs)) and v == vers]
            if hit:
                ctx_key = self._tc_keys[hit[0]][2]
                self._tc_keys.insert(0, self._tc_keys.pop(hit[0]))
            else:
                self._tc_counter += 1
                ctx_key = self._tc_counter
                self._tc_keys.insert(0, (tuple(ctxs), vers, ctx_key))
                del self._tc_keys[2:]                       # (the library keeps two keys)
        if use_graph or vace_ts is not None or t_frames is not None or nag is not None or slg or ctx_key:
            nv = 0 if vace_ts is None else len(vace_ts)
            for u in vace_ts or ():
                if tuple(u.shape) != (self.vace_in_dim, F, H, W):
                    raise _L.WanHipError(f"vace_context must be [{self.vace_in_dim},{F},{H},{W}], got {list(u.shape)}")
            VP = (c_void_p * nv)(*[u.data_ptr() for u in vace_ts]) if nv else None
            VS = (ctypes.c_float * nv)(*vace_scales) if nv else None
            a = _L.DitArgs(S, XP, tval, CP, ptr(yy), ptr(cos), ptr(sin), OP, F, H, W, ptr(ws), ws.numel(),
                           None if sp_struct is None else ctypes.cast(sp_struct, c_void_p), ctypes.cast(poll, c_void_p), None, FL, RP,
                           None, 1.0, t_frames, F if t_frames is not None else 0, nv, VP, VS,
                           *((0.0, 0.0, 0.0, None) if nag is None else (*nag, (ctypes.c_int * S)(*ctx_batches))),
                           (ctypes.c_int * len(slg))(*slg) if slg else None, len(slg), int(x_id), ctx_key)
            if use_graph:
                how = ctypes.c_int(0)
                rc = _L.load().wan_dit_forward_graph(self._ctx, ctypes.byref(a), stream_ptr(), ctypes.byref(how))
                self.last_graph_how = how.value
                outs = [o.clone() for o in so] if rc == 0 else outs
            else:
                rc = _L.load().wan_dit_forward_ex(self._ctx, ctypes.byref(a), stream_ptr())
        elif cache is None:
            rc = _L.load().wan_dit_forward(self._ctx, S, XP, tval, CP, ptr(yy), ptr(cos), ptr(sin), OP, F, H, W, ptr(ws),
                                           ws.numel(), sp_struct, poll, None, stream_ptr())
        else:
            rc = _L.load().wan_dit_forward_skip(self._ctx, S, XP, tval, CP, ptr(yy), ptr(cos), ptr(sin), OP, F, H, W, ptr(ws),
                                                ws.numel(), sp_struct, poll, None, FL, RP, stream_ptr())
        if rc == _L.WAN_ABORTED:
            return [None] * S                               # model.py:1997-1998
        check(rc, "wan_dit_forward")
        if sp is not None:
            outs = [sp.gather_output(o, (F, H // 2, W // 2)) for o in outs]
        return outs

    __call__ = forward
