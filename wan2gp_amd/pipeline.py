"""WanAny2VHIP -- the sampler loop of `WanAny2V.generate()` on resident HIP models.

Mirrors models/wan/any2video.py for the t2v / i2v2.2 path:
  scheduler build (:506-545) -> seed generator (:548-549) -> target_shape (:1166) ->
  RoPE tables (:1192) -> noise (:1470) -> per-step expert/guidance switch (:1437-1443,
  :1491-1492) -> joint CFG pass (:1626-1634) or two single passes (:1638-1643) ->
  CFG combine (:1722) -> scheduler.step (:1733) -> callback (:1743-1750) ->
  VAE decode to uint8 (:1763,:1784) -> {"x": ..., "latent_slice": None} (:1810-1826).
Text encoding (UMT5) is outside the hot path: the caller passes the already encoded
`context` / `context_null` ([1,512,4096], zero padded as :590 does), or a `text_encoder`
callable.  Returns None when interrupted, like the reference.
"""
from typing import Callable, Optional

import os

import torch

from .rope import get_rotary_pos_embed
from .schedulers import (EulerScheduler, FlowDPMSolverMultistepScheduler, FlowMatchScheduler, FlowUniPCMultistepScheduler,
                         HipScheduler, LCMScheduler, cfg_combine, get_sampling_sigmas, retrieve_timesteps)


# keyword -> the reference's default (any2video.py:414-503): a non-default value asks for something generate() below does not do.
# NOT in this table although wgp.py sets them on EVERY call (wgp.py:7762-7885): `causal_attention=True` (hard-coded at :7826; the
# reference's generate() does not even declare it -- it falls into **bbargs), `overlap_noise` (the UI's sliding-window default 20;
# read only for VACE with overlapped latents, served below), `overlapped_latents` (the latent slice of the previous sliding
# window, wgp.py's window loop: read by the reference on the VACE path -- any2video.py:1150-1152, :1524-1526, served below -- and on the
# svi_pro path, which does not reach this backend; the i2v path's use of it is switched off there, :779), `prefix_video` / `pre_video_frame` /
# `conditioning_latents_size` (non-empty whenever a start image or a source video is used, wgp.py:7378-7394, :7714; read only on
# the svi_pro / infinitetalk / scail2 / reference-image paths, any2video.py:659-728, :861-897, none of which reaches this backend).
_UNSERVED_WHEN_SET = {"input_faces": None,
                      "input_custom": None, "audio_scale": None, "audio_proj": None, "audio_context_lens": None, "audio_guide": None,
                      "audio_guide2": None, "input_waveform": None, "alt_guide_scale": 1.0,
                      "speakers_bboxes": None, "image_mode": 0, "face_arc_embeds": None, "vae_upsampler": None}
# (`control_scale_alt` is read for the lynx models only, any2video.py:1103-1104: accepted without effect like there)


def _same(v, default):
    try:
        return bool(v == default)
    except Exception:                                         # tensors and the like: set
        return False


def _resize_lanczos(img, h, w):
    """`resize_lanczos` (shared/utils/utils.py:341-347): [-1,1] float [3,H,W] -> 8-bit image (truncating cast) -> PIL Lanczos -> [-1,1]."""
    import numpy as np
    from PIL import Image
    a = (img + 1).float().mul_(127.5)
    pil = Image.fromarray(np.clip(a.movedim(0, -1).cpu().numpy(), 0, 255).astype(np.uint8))
    pil = pil.resize((w, h), resample=Image.Resampling.LANCZOS)
    return torch.from_numpy(np.array(pil).astype(np.float32)).movedim(-1, 0).div(127.5).sub_(1)


class _Progress:
    """The progress protocol `WanAny2V.generate` speaks to wgp.py's callback (any2video.py:1410-1411, :1434, :1442, :1446,
    :1743-1750): `(-1, None, True)` once the setup is done, `(-1, None, True, override_num_inference_steps=, denoising_extra=)`
    in front of the loop, `(step - 1, denoising_extra=)` when a guidance phase begins, `(step, preview latents, False,
    denoising_extra=)` behind every step.  The keyword half is passed only to callbacks whose signature takes it (and the phase
    notice only to those callable with the step alone), so plain `cb(step, latents, flag)` callables keep working."""

    def __init__(self, callback, set_header_text=None):
        self.callback, self.set_header_text, self.extra = callback, set_header_text, ""
        self.keywords = self.step_only = False
        if callback is not None:
            import inspect
            try:
                params = list(inspect.signature(callback).parameters.values())
                self.keywords = any(q.kind is q.VAR_KEYWORD for q in params) or \
                    {"override_num_inference_steps", "denoising_extra"} <= {q.name for q in params}
                # the phase notice is `callback(step - 1, denoising_extra=)`: only for callables whose other arguments are optional
                self.step_only = self.keywords and all(q.default is not q.empty or q.kind in (q.VAR_POSITIONAL, q.VAR_KEYWORD)
                                                       for q in params[1:])
            except (TypeError, ValueError):
                self.keywords = self.step_only = False

    def ready(self):
        if self.callback is not None:
            self.callback(-1, None, True)

    def begin(self, num_steps, guide_phases, two_experts, phases_description):
        if phases_description and self.set_header_text is not None:
            self.set_header_text(phases_description)
        if guide_phases > 1:
            self.extra = f"Phase 1/{guide_phases} High Noise" if two_experts else f"Phase 1/{guide_phases}"
        if self.callback is not None:
            if self.keywords:
                self.callback(-1, None, True, override_num_inference_steps=num_steps, denoising_extra=self.extra)
            else:
                self.callback(-1, None, True)

    def phase(self, step_no, phase_no, guide_phases, two_experts, low_noise):
        self.extra = f"Phase {phase_no}/{guide_phases}" + ((" Low Noise" if low_noise else " High Noise") if two_experts else "")
        if self.callback is not None and self.step_only:
            self.callback(step_no - 1, denoising_extra=self.extra)

    def step(self, i, preview):
        if self.callback is not None:
            if self.keywords:
                self.callback(i, preview, False, denoising_extra=self.extra)
            else:
                self.callback(i, preview, False)



class WanAny2VHIP:
    def __init__(self, model, model2=None, vae=None, text_encoder: Optional[Callable] = None, device="cuda",
                 num_train_timesteps=1000, vae_stride=(4, 8, 8), patch_size=(1, 2, 2)):
        self.model, self.model2, self.vae, self.text_encoder = model, model2, vae, text_encoder
        self.device = torch.device(device)
        self.num_train_timesteps = num_train_timesteps
        self.vae_stride, self.patch_size = vae_stride, patch_size
        self._interrupt = False
        # Wan2.1 i2v: the CLIP visual tower whose [n,257,1280] features the model's image branch takes.  It is the host application's own
        # `CLIPModel` (models/wan/modules/clip.py, any2video.py:127-132), handed over by the plugin's load_model: one 257-token ViT-H
        # forward per video, outside the denoise path.  None: generate() must be given clip_fea.
        self.clip, self.flf = None, False
        # multi-GPU: None, or an sp.CfgParallel -- the two CFG streams on the two halves of the world (its `attach()` has put the
        # half's sequence-parallel group on the experts); `model.sp` alone = sequence parallelism over the whole world
        self.cfg_parallel = None
        import collections
        self._text_cache, self._text_cache_bytes = collections.OrderedDict(), 0

    def _scheduler(self, sample_solver, sampling_steps, shift, native=True):
        """native=False: the Python mirrors even on a GPU -- their `timesteps` / `sigmas` tables can be cut short by the caller
        (video-to-video, any2video.py:1029-1033); the library's scheduler object owns its tables."""
        if native and sample_solver in ("unipc", "", "euler") and torch.device(self.device).type == "cuda":
            s = HipScheduler("euler" if sample_solver == "euler" else "unipc", num_train_timesteps=self.num_train_timesteps)
            s.set_timesteps(sampling_steps, device=self.device, shift=shift)          # wan_sched_* of the C ABI
        elif sample_solver == "euler":
            s = EulerScheduler(num_train_timesteps=self.num_train_timesteps, use_timestep_transform=True)
            s.set_timesteps(sampling_steps, device=self.device, shift=shift)
        elif sample_solver in ("unipc", ""):
            s = FlowUniPCMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                            use_dynamic_shifting=False)
            s.set_timesteps(sampling_steps, device=self.device, shift=shift)
        elif sample_solver == "dpm++":                                              # any2video.py:524-533
            s = FlowDPMSolverMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                use_dynamic_shifting=False)
            retrieve_timesteps(s, device=self.device, sigmas=get_sampling_sigmas(sampling_steps, shift))
        elif sample_solver == "causvid":                                            # :513-517
            s = FlowMatchScheduler(num_inference_steps=sampling_steps, shift=shift, sigma_min=0, extra_one_step=True)
            s.timesteps = torch.tensor([1000, 934, 862, 756, 603, 410, 250, 140, 74])[:sampling_steps].to(self.device)
            s.sigmas = torch.cat([s.timesteps / 1000, torch.tensor([0.], device=self.device)])
        elif sample_solver == "lcm":                                                # :534-543
            n = min(sampling_steps, 8)
            s = LCMScheduler(num_train_timesteps=self.num_train_timesteps, num_inference_steps=n, shift=shift)
            s.set_timesteps(n, device=self.device, shift=shift)
        else:
            raise NotImplementedError(f"Unsupported Scheduler {sample_solver}")
        return s, s.timesteps

    def build_i2v_conditioning(self, image_start, frame_num, height, width, VAE_tile_size=0, motion_amplitude=1.0, image_end=None,
                               add_frames_for_end_image=False):
        """y = cat(mask[4,f,h,w], vae.encode(known frames + zero frames)[16,f,h,w]) (any2video.py:699-774) and the clean
        latents of the known frames that are re-injected every step (:775-782).  `image_start`: one image [3,H,W] or a
        prefix video [3,P,H,W] to continue (control_video, :671-680); `motion_amplitude` > 1 stretches the latent
        differences to the first frame (:761-770).  `image_end` [3,H,W]: the clip's last frame is known as well (:684-704,
        :747-751): it closes the encoded video and its mask entry is 1; for the Wan2.1 i2v model (`add_frames_for_end_image`,
        :685-691) the clip grows by one frame -- one more latent frame, encoded without the causal cache (any_end_frame) and
        trimmed from the latents before they are decoded (:1759)."""
        dev = self.device
        video = image_start.to(device=dev, dtype=torch.float32)
        if video.dim() == 3:
            video = video.unsqueeze(1)                                                 # [3,1,H,W]
        P = video.shape[1]
        end = image_end is not None
        add = bool(end and add_frames_for_end_image)
        if add:
            frame_num += 1                                                             # :689
        lat_h, lat_w = height // self.vae_stride[1], width // self.vae_stride[2]
        if end:                                                                        # :700-705
            enc = torch.cat([video, torch.zeros(3, frame_num - P - 1, height, width, device=dev),
                             image_end.to(device=dev, dtype=torch.float32).unsqueeze(1)], dim=1)
        else:
            enc = torch.cat([video, torch.zeros(3, frame_num - P, height, width, device=dev)], dim=1)  # :739
        lat_y = self.vae.encode([enc], VAE_tile_size, any_end_frame=True)[0] if add else self.vae.encode([enc], VAE_tile_size)[0]   # :743
        msk = torch.ones(1, frame_num, lat_h, lat_w, device=dev)                                   # :746-757
        if end:
            msk[:, P:-1] = 0
        else:
            msk[:, P:] = 0
        if add:
            msk = torch.cat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:-1],
                             torch.repeat_interleave(msk[:, -1:], repeats=4, dim=1)], dim=1)
        else:
            msk = torch.cat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:]], dim=1)
        msk = msk.view(1, msk.shape[1] // 4, 4, lat_h, lat_w).transpose(1, 2)[0]
        if motion_amplitude > 1:                                                                   # :761-770
            base = lat_y[:, :1]
            diff = lat_y[:, P:] - base
            mean = diff.mean(dim=(0, 2, 3), keepdim=True)
            scaled = torch.clamp(base + (diff - mean) * motion_amplitude + mean, -6, 6)
            lat_y = torch.cat([lat_y[:, :P], scaled[:, :-1], lat_y[:, -1:]] if end else [lat_y[:, :P], scaled], dim=1)
        y = torch.cat([msk, lat_y.to(msk.dtype)])                                                  # :774
        n_known = int(1 + (P - 1) // 4)                                                            # :775
        return y, lat_y[:, :n_known].clone().unsqueeze(0)

    # ---- VACE context (any2video.py:270-331, :1128-1147) ---------------------------------------------------------------------
    def vace_encode_frames(self, frames, ref_images, masks=None, tile_size=0):
        """`WanAny2V.vace_encode_frames`: VAE latents of the control video, split by the mask into an inactive (kept) and a
        reactive (to be generated) half -> 32 channels; reference images, if any, are prepended as extra latent frames."""
        ref_images = [ref_images] * len(frames)
        if masks is None:
            latents = self.vae.encode(frames, tile_size=tile_size)
        else:
            inactive = self.vae.encode([i * (1 - m) + 0 * m for i, m in zip(frames, masks)], tile_size=tile_size)
            reactive = self.vae.encode([i * m + 0 * (1 - m) for i, m in zip(frames, masks)], tile_size=tile_size)
            latents = [torch.cat((u, c), dim=0) for u, c in zip(inactive, reactive)]
        out = []
        for latent, refs in zip(latents, ref_images):
            if refs is not None:
                ref_latent = self.vae.encode(refs, tile_size=tile_size)
                if masks is not None:
                    ref_latent = [torch.cat((u, torch.zeros_like(u)), dim=0) for u in ref_latent]
                assert all(x.shape[1] == 1 for x in ref_latent)
                latent = torch.cat([*ref_latent, latent], dim=1)
            out.append(latent)
        return out

    def vace_encode_masks(self, masks, ref_images=None):
        """`WanAny2V.vace_encode_masks`: the pixel mask folded 8x8 into 64 channels at latent resolution, nearest-exact
        resampled in time to the latent frame count; zero frames in front for reference images."""
        ref_images = [ref_images] * len(masks)
        st, sh, sw = self.vae_stride
        out = []
        for mask, refs in zip(masks, ref_images):
            _, depth, height, width = mask.shape
            new_depth = int((depth + 3) // st)
            height, width = 2 * (int(height) // (sh * 2)), 2 * (int(width) // (sw * 2))
            m = mask[0].view(depth, height, sh, width, sh).permute(2, 4, 0, 1, 3).reshape(sh * sw, depth, height, width)
            m = torch.nn.functional.interpolate(m.unsqueeze(0), size=(new_depth, height, width), mode="nearest-exact").squeeze(0)
            if refs is not None:
                m = torch.cat((torch.zeros(m.shape[0], len(refs), *m.shape[-2:], dtype=m.dtype, device=m.device), m), dim=1)
            out.append(m)
        return out

    def vace_context(self, input_frames, input_masks, input_ref_images=None, tile_size=0, input_ref_masks=None):
        """z = [cat(z0, m0)] as built at any2video.py:1136-1146: [96, F, H, W] per control video.  With a background mask for the first
        reference image (`input_ref_masks[0]`, [1,1,H,W]: an outpainted background, :1138-1145) that image's latent frame and mask frame
        are replaced by the masked encodings (inactive / reactive halves, folded mask) in every context."""
        z0 = self.vace_encode_frames(input_frames, input_ref_images, masks=input_masks, tile_size=tile_size)
        m0 = self.vace_encode_masks(input_masks, input_ref_images)
        if input_ref_masks is not None and len(input_ref_masks) > 0 and input_ref_masks[0] is not None:
            zbg = self.vace_encode_frames(input_ref_images[:1] * len(input_frames), None, masks=input_ref_masks[0], tile_size=tile_size)
            mbg = self.vace_encode_masks(input_ref_masks[:1] * len(input_frames), None)
            for zz0, mm0, zzbg, mmbg in zip(z0, m0, zbg, mbg):
                zz0[:, 0:1] = zzbg
                mm0[:, 0:1] = mmbg
        return [torch.cat([zz, mm.to(zz.dtype)], dim=0) for zz, mm in zip(z0, m0)]

    def clip_features(self, image_start, image_end=None):
        """any2video.py:945-954: start (and, for flf2v_720p, end) image [3,H,W] in [-1,1] -> Lanczos resize to the tower's input size
        (through 8-bit PIL images, `resize_lanczos`, shared/utils/utils.py:341-347) -> `clip.visual([[3,1,S,S], ...])` = [n,257,1280]."""
        size = self.clip.model.image_size
        start = _resize_lanczos(image_start, size, size)
        if self.flf or "img_emb.emb_pos" in getattr(self.model, "_weights", ()):
            end = _resize_lanczos(image_end, size, size) if image_end is not None else start
            return self.clip.visual([start[:, None, :, :], end[:, None, :, :]])
        return self.clip.visual([start[:, None, :, :]])

    def get_loras_transformer(self, get_model_recursive_prop, base_model_type, model_type, video_prompt_type, model_mode, **kwargs):
        """wgp.py asks the pipeline for model-specific preloaded LoRAs (any2video.py:1828-1839): only the `animate` and
        `vace_ditto_14B` model types have any, neither of which this backend serves -> none."""
        return [], []

    # ---- multi-GPU: what must be bit-identical on every rank -------------------------------------------------------------------
    def _ranks_share_latents(self):
        """True when this process is one rank of a CFG-parallel / sequence-parallel world: latents are replicated and nothing is
        broadcast per step (sp.py), so every random draw of generate() has to come out the same on every rank."""
        if getattr(self, "cfg_parallel", None) is not None:
            return True
        sp = getattr(self.model, "sp", None)
        return sp is not None and getattr(sp, "world", 1) > 1

    def _latent_group(self):
        """(process group, global rank of its first member) of the ranks that hold the same latents as this one, or None outside an
        initialised multi-rank world.  CFG parallelism spans the ranks 0 .. world - 1 of the default group (sp.py CfgParallel builds its
        halves and pairs from them): the default group, source 0.  Plain sequence parallelism: the group the SequenceParallel object
        was given (None = the default group) -- which may be a sub-group of a larger world, so the source is translated to a GLOBAL rank
        (round-4 advisor: a broadcast on WORLD with src 0 hangs for a sub-group whose first member is not global rank 0)."""
        if not self._ranks_share_latents():
            return None
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if getattr(self, "cfg_parallel", None) is not None:
            return None, 0
        grp = getattr(self.model.sp, "group", None)
        return grp, (dist.get_global_rank(grp, 0) if grp is not None else 0)

    def _replicated_seed(self, seed):
        """seed >= 0 as given; a negative seed means "draw one" -- in a multi-rank world the draw of the latent-sharing group's first
        rank, for all of them.  Drawn from the OS (not torch.seed(), which would also re-seed this rank's global generator)."""
        if seed >= 0:
            return seed
        drawn = int.from_bytes(os.urandom(4), "little") % (2 ** 31)
        lg = self._latent_group()
        if lg is not None:
            import torch.distributed as dist
            grp, src = lg
            t = torch.tensor([drawn], dtype=torch.int64, device=self.device if dist.get_backend(grp) != "gloo" else "cpu")
            dist.broadcast(t, src=src, group=grp)
            drawn = int(t.item())
        return drawn

    def _replicated_randn_like(self, ref):
        """torch.randn_like(ref) from the process-global generator, as the reference draws the per-step noise of a pinned prefix
        (any2video.py:1517-1526).  That generator is neither seeded from `seed` nor the same on two ranks: in a multi-rank world rank
        0 draws and every rank takes its tensor (one broadcast of the prefix's size per step) -- otherwise the conditional and the
        unconditional half of a CFG-parallel world, or the shards of a sequence-parallel one, would denoise different latents."""
        noise = torch.randn_like(ref)
        lg = self._latent_group()
        if lg is not None:
            import torch.distributed as dist
            grp, src = lg
            if dist.get_backend(grp) == "gloo" and noise.is_cuda:
                host = noise.cpu()
                dist.broadcast(host, src=src, group=grp)
                noise = host.to(ref.device)
            else:
                noise = noise.contiguous()
                dist.broadcast(noise, src=src, group=grp)
        return noise

    def generate(self, input_prompt=None, n_prompt="", context=None, context_null=None, width=1280, height=720,
                 frame_num=81, batch_size=1, shift=5.0, sample_solver="unipc", sampling_steps=30, guide_scale=5.0,
                 guide2_scale=5.0, guide3_scale=5.0, switch_threshold=0, guide_phases=1, model_switch_phase=1, seed=-1, callback=None,
                 joint_pass=True, y=None, image_start=None, latents=None, VAE_tile_size=0, return_latents=False,
                 loras_slists=None, switch2_threshold=0, enable_RIFLEx=False, cfg_star_switch=0, cfg_zero_step=5, apg_switch=0,
                 input_frames=None, input_masks=None, context_scale=None, sub_parallel_window_size=0, sub_parallel_window_overlap=0,
                 motion_amplitude=1.0, clip_fea=None, input_video=None, NAG_scale=0, NAG_tau=3.5, NAG_alpha=0.5, image_end=None,
                 return_latent_slice=None, video_prompt_type="", denoising_strength=1.0, masking_strength=1.0, keep_frames_parsed=None,
                 prefix_frames_count=0, self_refiner_setting=0, self_refiner_plan="", self_refiner_f_uncertainty=0.0,
                 self_refiner_certain_percentage=0.999, perturbation_layers=None, perturbation_start=0.0, perturbation_end=1.0, set_header_text=None,
                 overlapped_latents=None, overlap_noise=0, input_ref_images=None, input_ref_masks=None, input_frames2=None, input_masks2=None,
                 color_correction_strength=1, window_start_frame_no=0, **bbargs):
        if batch_size != 1:
            raise NotImplementedError("batch_size 1 per generate() call (as wgp.py drives it)")
        # wgp.py hands every generate() the union of all variants' keywords (wgp.py:7762-7885); the ones below change the result
        # when they are set and belong to paths this backend does not serve: refuse instead of producing a different video
        # (the other unknown keywords -- UI handles, file names, progress hooks -- are swallowed like the reference's **bbargs)
        unserved = {k: bbargs[k] for k, default in _UNSERVED_WHEN_SET.items()
                    if bbargs.get(k, None) is not None and not _same(bbargs[k], default)}
        if unserved:
            raise NotImplementedError(f"WanAny2VHIP.generate: {sorted(unserved)} select reference paths outside this backend "
                                      "(faces / audio / image outputs / VAE upsampler)")
        if context is None:
            if self.text_encoder is None or input_prompt is None:
                raise ValueError("pass `context`/`context_null` ([1,512,4096] bf16) or a text_encoder + input_prompt")
            # any2video.py:587-593: the encoder returns unpadded [n_tokens, 4096]; zero-pad to text_len and add the batch axis
            text_len = getattr(self.model, "text_len", 512)

            def _encode(prompt):
                # `TextEncoderCache.encode` (shared/utils/text_encoder_cache.py:16-61; any2video.py:589, :592): the encoder's output per
                # prompt string, least recently used first out beyond 100 MB -- every sliding window repeats the video's prompts
                c = self._text_cache.get(prompt)
                if c is None:
                    c = self.text_encoder([prompt], self.device)[0]
                    self._text_cache[prompt] = c
                    self._text_cache_bytes += c.numel() * c.element_size()
                    while self._text_cache_bytes > 100 * 1024 * 1024 and len(self._text_cache) > 1:
                        _, old = self._text_cache.popitem(last=False)
                        self._text_cache_bytes -= old.numel() * old.element_size()
                else:
                    self._text_cache.move_to_end(prompt)
                c = c.to(device=self.device, dtype=torch.bfloat16)
                return torch.cat([c, c.new_zeros(text_len - c.size(0), c.size(1))]).unsqueeze(0)
            context = _encode(input_prompt)
            context_null = _encode(n_prompt)
        # normalized attention guidance (any2video.py:607-608): the positive prompt's context carries the negative one as a second
        # batch entry; the text cross-attention of every block combines the two results (model.py:260-292)
        nag = (float(NAG_scale), float(NAG_tau), float(NAG_alpha)) if NAG_scale > 1 else None
        for m in (self.model, self.model2):
            if m is not None and hasattr(m, "nag"):
                m.nag = nag
        if nag is not None:
            if context_null is None:
                raise ValueError("NAG_scale > 1 needs the negative prompt's context (context_null)")
            context = torch.cat([context, context_null.to(context)], dim=0)
        # wgp.py hands `image_start` / `image_end` to EVERY model (wgp.py:7765-7766); the reference reads them on its i2v path only
        # (any2video.py:651-785) -- a 5B or VACE generation with a start image is conditioned through input_video / input_frames
        mt = getattr(self.model, "model_type", None)
        if mt is not None and mt not in ("i2v", "i2v2_2"):
            image_start = image_end = None
        black_start = False
        if mt is not None and mt in ("i2v", "i2v2_2") and input_video is None and image_start is None and y is None:
            input_video = torch.full((3, 1, height, width), -1.0)                     # :667-669: no start image -> a black frame ...
            black_start = True                                                        # ... and no colour matching against it (:669)
        if mt == "i2v" and clip_fea is None:
            src = input_video if input_video is not None else image_start
            if self.clip is None or src is None:
                raise ValueError("a Wan2.1 i2v model (model_type 'i2v') needs clip_fea [1,257,1280] -- the CLIP vision features of the start "
                                 "image (any2video.py:945-954) -- or a pipeline with the host application's CLIP tower (self.clip) and a start image")
            clip_fea = self.clip_features(src[:, -1] if src.dim() == 4 else src, image_end)
        dev = self.device
        if input_video is not None:
            height, width = input_video.shape[-2:]                                    # any2video.py:571: the video to continue sets the size
        elif getattr(self.model, "model_type", None) == "ti2v2_2":
            height, width = (height // 32) * 32, (width // 32) * 32                   # :1063-1065: the 5B model's VAE stride 16 x patch 2
        # video-to-video ("G" in video_prompt_type, any2video.py:1004-1044): start from the VAE latents of `input_frames`
        v2v_on = "G" in (video_prompt_type or "") and input_frames is not None
        if not v2v_on:
            denoising_strength = 1                                                      # :1044
        # self-refining steps (any2video.py:1485-1488): the handler copies and restores the scheduler -> a Python mirror
        refiner_handler = None
        if self_refiner_setting > 0:
            from . import refiner
            refiner_handler = refiner.create(self_refiner_plan, self_refiner_f_uncertainty, self_refiner_setting, self_refiner_certain_percentage)
        sample_scheduler, timesteps = self._scheduler(sample_solver, sampling_steps, shift,
                                                      native=not ((v2v_on and denoising_strength < 1) or refiner_handler is not None))
        seed_g = torch.Generator(device=dev)
        seed_g.manual_seed(self._replicated_seed(seed))
        lat_frames = (frame_num - 1) // self.vae_stride[0] + 1                       # any2video.py:647
        # start + end image (any2video.py:684-691): the Wan2.1 i2v model gets one extra frame -- one more latent frame, encoded
        # without the causal cache and trimmed from the latents before decoding; Wan2.2 i2v keeps the frame count
        trim_frames = 0
        add_end = image_end is not None and getattr(self.model, "model_type", None) == "i2v"
        if image_end is not None and image_start is None and input_video is None:    # wgp.py's spelling of the start is input_video
            raise ValueError("image_end needs image_start (any2video.py:667-704)")
        if add_end:
            lat_frames = int((frame_num + 1 - 2) // self.vae_stride[0] + 2)
            trim_frames = 1
        # VACE reference images (any2video.py:1128-1149, ref_images_before): each image becomes one extra latent frame IN FRONT of
        # the control context and of the latents (target_shape :1166); those frames are cut off before decoding (:1758) and from the
        # previews (:1745).  Reference images of other model families (phantom, lynx, ...) are not served.
        ref_count = 0
        if input_ref_images is not None and len(input_ref_images) > 0:
            if getattr(self.model, "vace_layers", None) is None or input_frames is None:
                raise NotImplementedError("WanAny2VHIP.generate: input_ref_images are served on the VACE path only (a model with VACE "
                                          "blocks and a control video)")
            ref_count = len(input_ref_images)
        target_shape = (getattr(self.model, "out_dim", 16), lat_frames + ref_count, height // self.vae_stride[1],
                        width // self.vae_stride[2])                                   # :1166 (48 channels, stride 16 for ti2v 5B)
        freqs = get_rotary_pos_embed(target_shape[1:], enable_RIFLEx=bool(enable_RIFLEx), device=dev)   # :1192
        if latents is None:
            latents = torch.randn(batch_size, *target_shape, dtype=torch.float32, device=dev, generator=seed_g)  # :1470
        else:
            latents = latents.to(device=dev, dtype=torch.float32).clone()
        # ---- image2video conditioning (any2video.py:651-785, plain i2v2_2: one start image) -----------
        # The reference's i2v path takes its conditioning from `input_video` (any2video.py:671-680: control_video = input_video,
        # image_start = input_video[:, -1]); wgp.py always passes input_video = the start image as [3,1,H,W] or the video to
        # continue.  `image_start` (what direct callers of this class pass) is the same thing under the other name.
        ext_latents = None
        # the frame a later sliding window's colours are matched to after decoding (any2video.py:552, :1783-1808; color.py)
        color_reference_frame = None
        if getattr(self.model, "model_type", None) in ("i2v", "i2v2_2"):
            if input_video is None or image_end is not None or black_start:
                color_correction_strength = 0                                        # :667-669, :686-687
            if input_video is not None:
                image_start, input_video = input_video, None
                color_reference_frame = (image_start[:, -1] if image_start.dim() == 4 else image_start).unsqueeze(1).clone()   # :682
        if image_start is not None:
            if self.vae is None:
                raise ValueError("image_start needs a VAE to encode the conditioning video")
            y, ext_latents = self.build_i2v_conditioning(image_start, frame_num, height, width, VAE_tile_size, motion_amplitude,
                                                         image_end=image_end, add_frames_for_end_image=add_end)
        # ti2v (Wan2.2 5B) image / video conditioning by timestep injection (any2video.py:1060-1072, :1496-1499, :1753-1754):
        # the VAE latents of the source frames replace the first latent frames before every step and after the last one, and
        # those frames are given timestep 0 -- a per-frame t vector
        source_latents = None
        if input_video is not None and getattr(self.model, "model_type", None) != "ti2v2_2":
            # any2video.py:571 is the ONLY line that reads input_video for a t2v-class / VACE model -- height and width, taken above.
            # wgp.py passes it to every model type for every sliding window after the first (wgp.py:7740, :7995: input_video =
            # pre_video_guide, the overlap frames of the previous window); the VACE path takes those frames from input_frames /
            # overlapped_latents instead (any2video.py:837, :1150-1163)
            input_video = None
        if input_video is not None:
            if self.vae is None:
                raise ValueError("input_video (timestep injection, the ti2v_2_2 conditioning path) needs the Wan2.2 VAE")
            source_latents = self.vae.encode([input_video.to(dev)], VAE_tile_size)[0].unsqueeze(0)
        v2v, v2v_src, randn, start_step_no = None, None, None, 0
        original_timesteps = timesteps            # any2video.py:546: the whole schedule, also when video-to-video cuts it short
        if v2v_on:
            from . import video2video
            if self.vae is None:
                raise ValueError("video-to-video needs a VAE to encode input_frames")
            if tuple(input_frames.shape[-2:]) != (height, width):
                raise ValueError(f"input_frames are {tuple(input_frames.shape[-2:])}, height x width = {(height, width)} (any2video.py:1005)")
            # :1006 `self.vae.encode([input_frames])`: no tile size given -> WanVAE.encode's default of 256 (vae.py:1003), i.e. the
            # reference ALWAYS encodes the source in 256-pixel tiles, whatever VAE_tile_size says
            v2v_src = self.vae.encode([input_frames.to(dev)], 256)[0].unsqueeze(0)
            v2v = video2video.plan(input_frames, None if input_masks is None else input_masks.to(dev), v2v_src, lat_frames, sampling_steps,
                                   denoising_strength, masking_strength, list(keep_frames_parsed or []), prefix_frames_count, timesteps,
                                   sample_scheduler, device=dev, video_prompt_type=video_prompt_type)
            timesteps, start_step_no = v2v.timesteps, v2v.start_step_no
            if denoising_strength < 1:
                color_reference_frame = input_frames[:, -1:].clone()                              # :1008-1009
            randn = latents                                                                  # :1475 -- the SAME tensor, as there
        vace_kwargs, vace_overlap = {}, False
        if input_frames2 is not None and (input_frames is None or getattr(self.model, "vace_layers", None) is None):
            raise NotImplementedError("WanAny2VHIP.generate: input_frames2 is served as VACE's second control video only")
        if input_frames is not None and (not v2v_on or getattr(self.model, "vace_layers", None) is not None):
            # VACE control video + mask (any2video.py:1128-1147), reference images in front if given
            if self.vae is None or input_masks is None:
                raise ValueError("VACE needs a VAE, input_frames [3,T,H,W] and input_masks [1,T,H,W]")
            # a second control video (any2video.py:1129-1130): one more context, run through the same context blocks with its own scale
            if (input_frames2 is None) != (input_masks2 is None):
                raise ValueError("input_frames2 and input_masks2 come together (any2video.py:1129-1130)")
            z = self.vace_context([input_frames.to(dev)] + ([] if input_frames2 is None else [input_frames2.to(dev)]),
                                  [input_masks.to(dev)] + ([] if input_masks2 is None else [input_masks2.to(dev)]),
                                  [u.to(dev) for u in input_ref_images] if ref_count else None, VAE_tile_size,
                                  None if not ref_count or input_ref_masks is None else [None if u is None else u.to(dev) for u in input_ref_masks])
            if ref_count and input_ref_masks is not None and len(input_ref_masks) > 0 and input_ref_masks[0] is not None:
                color_reference_frame = input_ref_images[0].clone()                                  # :1139
            vace_kwargs = {"vace_context": z, "vace_context_scale": context_scale if context_scale is not None else [1.0] * len(z)}
            # sliding windows (any2video.py:1150-1152): wgp.py hands the previous window's last latent frames; the INACTIVE half of
            # the control video's first latent frames (the overlap, which the control video repeats) is what gets pinned: injected
            # re-noised in front of every step and clean behind the last one, like the i2v prefix
            vace_overlap = overlapped_latents is not None
            if vace_overlap:
                ext_latents = z[0][:16, :overlapped_latents.shape[2] + ref_count].clone().unsqueeze(0)      # :1151-1152
            if prefix_frames_count > 0:
                color_reference_frame = input_frames[:, prefix_frames_count - 1:prefix_frames_count].clone()   # :1153-1154
        any_guidance = guide_scale != 1 or (guide_phases > 1 and guide2_scale != 1)
        trans = self.model
        guidance_switch_done = guidance_switch2_done = False
        text_momentum = None
        # LoRA multipliers per step (any2video.py:1431-1445, :1493): the reference re-selects the active multipliers on
        # every step (offload.set_step_no_for_lora); merged adapters are re-merged only when a step's multipliers change
        from .lora import get_model_switch_steps
        phase_switch_step, phase_switch_step2, phases_description = get_model_switch_steps(
            [float(t) for t in original_timesteps], guide_phases, 0 if self.model2 is None else model_switch_phase, switch_threshold,
            switch2_threshold)
        progress = _Progress(callback, set_header_text)
        # sub-parallel temporal windows (any2video.py:1199-1223, :1392-1397): per-step forwards on overlapping windows of latent
        # frames; step-skipping caches are parked while they are active (their residuals have the full clip's token count)
        from . import subparallel
        sub_win, sub_overlap = subparallel.window_latent_counts(sub_parallel_window_size, sub_parallel_window_overlap, lat_frames,
                                                                self.vae_stride[0])
        sub_windows = subparallel.build_windows(lat_frames, sub_win, sub_overlap)
        parked = []
        if sub_windows is not None:
            for m in (self.model, self.model2):
                if m is not None and getattr(m, "cache", None) is not None:
                    parked.append((m, m.cache))
                    m.cache = None

        def restore_caches():                                # clear() (any2video.py:1448-1462)
            for m, c in parked:
                c.previous_residual = None
                c.previous_modulated_input = None
                m.cache = c
            return None
        # multi-GPU: `self.cfg_parallel` (sp.CfgParallel) puts the conditional and the unconditional stream of every guided step on the
        # two halves of the world.  A step-skipping cache decides for the unconditional stream from the conditional stream's decision of
        # the same step (skipcache.decide, x_id 0 then 1); that decision never reads latents, so the unconditional rank makes it itself
        # (WanModelHIP.forward, cfg_parallel_stream) -- refused until round 6.
        cfg_parallel = getattr(self, "cfg_parallel", None)
        # step-skipping caches (any2video.py:1398-1408): reset, then pick the threshold that meets cache.multiplier
        # The reference configures only self.model.cache (any2video.py:1396-1406; wgp.py hands the SAME object to both
        # experts): one reset, threshold from model's time embedding.  A distinct cache object on model2 gets its own setup.
        seen = []
        for m in (self.model, self.model2):
            cache = getattr(m, "cache", None) if m is not None else None
            if cache is not None and not any(cache is c for c in seen):
                seen.append(cache)
                from . import skipcache
                skipcache.reset_for_generation(cache, 2)
                cache.num_steps = len(original_timesteps)
                cache.previous_modulated_input = None
                if cache.cache_type == "tea":
                    m.compute_teacache_threshold(max(cache.start_step, start_step_no), original_timesteps, cache.multiplier)
                else:
                    m.compute_magcache_threshold(max(cache.start_step, start_step_no), original_timesteps, cache.multiplier)
        progress.ready()                                                                           # :1410-1411
        progress.begin(len(timesteps), guide_phases, self.model2 is not None, phases_description)  # :1434-1436, :1446
        kwargs = {"freqs": freqs, "pipeline": self, "callback": callback, "y": y, "max_steps": len(timesteps), **vace_kwargs}
        if clip_fea is not None:
            kwargs["clip_fea"] = clip_fea
        try:
            for i, t in enumerate(timesteps):
                # update_guidance (:1437-1443): phase 2 begins once t <= switch_threshold
                if guide_phases >= 2 and not guidance_switch_done and t <= switch_threshold:
                    if model_switch_phase == 1 and self.model2 is not None:
                        trans = self.model2
                    guide_scale, guidance_switch_done = guide2_scale, True
                    progress.phase(i, 2, guide_phases, self.model2 is not None, trans is self.model2)
                if guide_phases >= 3 and not guidance_switch2_done and t <= switch2_threshold:          # phase 3 (:1492)
                    if model_switch_phase == 2 and self.model2 is not None:
                        trans = self.model2
                    guide_scale, guidance_switch2_done = guide3_scale, True
                    progress.phase(i, 3, guide_phases, self.model2 is not None, trans is self.model2)
                timestep = torch.stack([t])
                if source_latents is not None:                   # any2video.py:1496-1499
                    n_src = source_latents.shape[2]
                    latents[:, :, :n_src] = source_latents
                    timestep = torch.full((target_shape[1],), int(t), dtype=torch.int64, device=latents.device)
                    timestep[:n_src] = 0
                kwargs.update({"t": timestep, "current_step_no": i, "real_step_no": start_step_no + i})
                # skip-layer guidance inside its window of steps (any2video.py:1502)
                kwargs["perturbation_layers"] = perturbation_layers if (perturbation_layers is not None and int(perturbation_start * sampling_steps)
                                                                        <= i < int(perturbation_end * sampling_steps)) else None
                if v2v is not None:                              # any2video.py:1504-1515: the noised source in front of the first steps
                    latents = video2video.inject(latents, randn, v2v_src, t, i, denoising_strength, v2v)
                if loras_slists is not None and getattr(trans, "loras", None) is not None:
                    trans.loras.set_step(loras_slists, len(original_timesteps), start_step_no + i, phase_switch_step, phase_switch_step2)   # :1444, :1493
                if ext_latents is not None:                      # any2video.py:1517-1523: re-noise the known first latent
                    f = float(t) / 1000.0
                    n = ext_latents.shape[2]
                    latents[:, :, :n] = ext_latents * (1.0 - f) + self._replicated_randn_like(ext_latents) * f
                    if vace_overlap:                             # :1523-1526: the context's overlap frames get `overlap_noise` / 1000 of noise
                        of = overlap_noise / 1000
                        for zz in vace_kwargs["vace_context"]:
                            zz[0:16, ref_count:n] = ext_latents[0, :, ref_count:] * (1.0 - of) + self._replicated_randn_like(ext_latents[0, :, ref_count:]) * of
                def denoise_with_cfg(lat):                       # denoise_with_cfg_fn, plain two-stream branch (any2video.py:1610-1722)
                    nonlocal text_momentum
                    if guide_scale == 1 or not any_guidance:
                        ret = trans(x=[lat], context=[context], **kwargs)
                        return None if (self._interrupt or ret[0] is None) else ret[0]
                    if cfg_parallel is not None:
                        # the two streams on the two halves of the world, swapped once per step (sp.CfgParallel): every rank ends up
                        # with the (cond, uncond) pair the joint pass returns
                        ret = cfg_parallel.guided_pair(trans, lat, context, context_null, **kwargs)
                        if self._interrupt or ret is None:
                            return None
                    elif joint_pass:
                        ret = trans(x=[lat, lat], context=[context, context_null], **kwargs)              # :1626-1634
                        if self._interrupt or ret[0] is None:
                            return None
                    else:
                        ret = []
                        for x_id, c in enumerate((context, context_null)):                               # :1638-1643
                            r = trans(x=[lat], context=[c], x_id=x_id, **kwargs)[0]
                            if self._interrupt or r is None:
                                return None
                            ret.append(r)
                    if apg_switch != 0 or cfg_star_switch:
                        # adaptive projected guidance / CFG-Zero* (:1703-1721; momentum -0.75, norm threshold 55, :1476-1478)
                        from . import guidance
                        if apg_switch != 0 and text_momentum is None:
                            text_momentum = guidance.MomentumBuffer(-0.75)
                        return guidance.combine(ret[0], ret[1], float(guide_scale), i, apg_switch, cfg_star_switch, cfg_zero_step,
                                                text_momentum, 55)
                    return cfg_combine(ret[0], ret[1], float(guide_scale))                              # :1722

                if sub_windows is not None:                      # any2video.py:1724: one forward per temporal window, blended
                    from . import subparallel

                    def denoise_fn(lat):
                        return subparallel.denoise(lat, denoise_with_cfg, sub_windows, sub_overlap, kwargs,
                                                   (target_shape[2] // self.patch_size[1]) * (target_shape[3] // self.patch_size[2]),
                                                   prefix=ref_count)          # :1222: reference-image frames lead every window
                else:
                    denoise_fn = denoise_with_cfg
                noise_pred = denoise_fn(latents)
                if noise_pred is None:
                    return None
                if refiner_handler is not None:                  # :1729-1731: the scheduler step, repeated on the plan's steps
                    sk = {} if isinstance(sample_scheduler, FlowMatchScheduler) else {"generator": seed_g}
                    latents, sample_scheduler = refiner_handler.step(i, latents, noise_pred, t, timesteps, target_shape, seed_g,
                                                                     sample_scheduler, sk, denoise_fn)
                    if latents is None:
                        return None
                elif isinstance(sample_scheduler, FlowMatchScheduler):                                  # :1463-1467
                    latents = sample_scheduler.step(noise_pred[:, :, :target_shape[1]], t, latents)[0]
                else:
                    latents = sample_scheduler.step(noise_pred[:, :, :target_shape[1]], t, latents, generator=seed_g)[0]
                if v2v is not None:                              # :1737-1740: outside the mask, the source at the next step's noise level
                    latents = video2video.merge(latents, randn, v2v_src, timesteps, i, v2v)
                if callback is not None:                         # :1743-1750: the preview leaves out the padded end-image frame
                    pv = latents[:, :, ref_count:] if ref_count else latents                      # :1745: not the reference-image frames
                    progress.step(i, (pv[:, :, :-trim_frames] if trim_frames > 0 else pv)[0])
        finally:
            restore_caches()                                 # also when a forward raises: parked caches must come back
        if source_latents is not None:
            latents[:, :, :source_latents.shape[2]] = source_latents                               # :1753-1754
        if ext_latents is not None:
            latents[:, :, :ext_latents.shape[2]] = ext_latents                                     # :1755-1756
        if ref_count:
            latents = latents[:, :, ref_count:]                                                    # :1758 (ref_images_before)
        if trim_frames > 0:
            latents = latents[:, :, :-trim_frames]                                                 # :1759
        # :1760-1761: the latent frames a sliding-window caller asks back (a slice object over the latent time axis)
        latent_slice = latents[:, :, return_latent_slice].clone() if return_latent_slice is not None else None
        if return_latents or self.vae is None:
            return {"x": None, "latents": latents, "latent_slice": latent_slice}
        if getattr(self.vae, "sp", None) is None and getattr(self.model, "sp", None) is not None:
            self.vae.sp = self.model.sp                      # multi-GPU: a tiled decode spreads its tiles over the sequence-parallel ranks
        x0 = latents.unbind(0)                                                                     # :1763
        videos = self.vae.decode_to_cpu_uint8(x0, VAE_tile_size)[0]                                # :1784, :1798
        # :1783, :1799-1808: every window after the first is colour-matched (Lab mean / std per frame) to its reference frame
        if color_correction_strength > 0 and (window_start_frame_no + prefix_frames_count) > 1 and color_reference_frame is not None:
            from .color import correct_window
            videos = correct_window(videos, color_reference_frame, color_correction_strength)
        return {"x": videos, "latents": latents, "latent_slice": latent_slice}
