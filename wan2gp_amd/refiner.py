"""Self-refining sampler steps ("PnP" plug-and-play refinement) around the scheduler step of the sampler loop.

Mirrors what `WanAny2V.generate` does when `self_refiner_setting > 0` (models/wan/any2video.py:1485-1488, :1729-1730) with the
handler of shared/utils/self_refiner.py (`create_self_refiner_handler` :353-372, `PnPHandler` :146-351): on the steps a plan
names, the step is repeated up to `steps` times -- each repetition re-noises the current x0 estimate to the step's sigma, asks the
model again and takes the scheduler step from a restored scheduler state -- while positions whose x0 estimate stopped moving
(p-norm over the channel axis below a threshold) are frozen to their previous result; the repetitions stop early once almost every
position is frozen.

Host logic + latent-sized torch arithmetic; the model call is the caller's `denoise` function.  Pinned against the reference's own
handler on identical inputs by tests/test_refiner_vs_reference.py.  The scheduler has to be copyable (`copy.deepcopy`): generate()
uses the Python scheduler mirrors when a refiner is active.
"""
import copy
from typing import List, Optional, Tuple

import torch

DEFAULT_PLAN = [{"start": 1, "end": 5, "steps": 3}, {"start": 6, "end": 13, "steps": 1}]     # self_refiner.py:361-365


def parse_plan(text) -> Tuple[List[List[dict]], str]:
    """`normalize_self_refiner_plan` (:66-100): "start-end:steps, ..." rules, several plans separated by ';' (or an already
    parsed list of rule dicts).  Returns (plans, error message)."""
    if text is None:
        return [[]], ""
    if isinstance(text, list):
        return [[r for r in text if isinstance(r, dict) and "start" in r and "end" in r]], ""
    text = str(text).strip()
    if not text:
        return [[]], ""
    plans = []
    for segment in (s.strip() for s in text.split(";")):
        rules = []
        for chunk in (c.strip() for c in segment.split(",")):
            if not chunk:
                continue
            if ":" not in chunk:
                return [], f"Invalid format in '{chunk}'. Entries must be in 'start-end:steps' format."
            span, steps = (p.strip() for p in chunk.split(":", 1))
            if not steps:
                return [], f"Missing step count in '{chunk}'."
            first, last = (p.strip() for p in span.split("-", 1)) if "-" in span else (span, span)
            if not (_is_int(first) and _is_int(last)):
                return [], f"Range '{span}' must contain integers."
            if not _is_int(steps):
                return [], f"Steps '{steps}' must be an integer."
            rules.append({"start": int(first), "end": int(last), "steps": int(steps)})
        rules.sort(key=lambda r: r["start"])
        plans.append(rules)
    return plans, ""


def _is_int(s: str) -> bool:
    try:
        int(s)
        return True
    except ValueError:
        return False


class SelfRefiner:
    def __init__(self, plan=None, f_uncertainty: float = 0.0, p_norm=1, certain_percentage: float = 0.999, channel_dim: int = 1):
        plans, _ = parse_plan(plan)
        rules = plans[0] if plans and plans[0] else DEFAULT_PLAN
        self.repeats = {}                                             # step index -> repetitions (`_build_stochastic_step_map`)
        for r in rules:
            if isinstance(r, dict):
                a, b, n = r.get("start", r.get("begin")), r.get("end", r.get("stop")), r.get("steps", r.get("anneal", r.get("num_anneal_steps", 1)))
            elif isinstance(r, (list, tuple)):
                a, b, n = r[0], r[1], r[2]
            else:
                continue
            if int(n) > 0:
                for i in range(int(a), int(b) + 1):
                    self.repeats[i] = int(n)
        self.f_uncertainty, self.p_norm, self.certain_percentage, self.channel_dim = f_uncertainty, p_norm, certain_percentage, channel_dim
        self.history: List[Optional[tuple]] = [None]                  # (frozen mask | None, x0 estimate, next latents) per repetition
        self.settled = False

    # ---- one repetition's bookkeeping (`process_step`, :184-216) ----------------------------------------------------------------
    def _absorb(self, latents, next_latents, x0):
        prev = self.history[-1]
        mask = None
        if prev is not None:
            dim = self.channel_dim + latents.ndim if self.channel_dim < 0 else self.channel_dim
            moved = torch.norm(x0 - prev[1], p=self.p_norm, dim=dim) / latents.shape[dim]
            mask = moved < self.f_uncertainty
            if prev[0] is not None:
                mask = mask | prev[0]
            if mask.sum() / mask.numel() > self.certain_percentage:
                self.settled = True
            w = mask.to(latents.dtype).unsqueeze(dim)
            next_latents = w * prev[2] + (1.0 - w) * next_latents
            x0 = w * prev[1] + (1.0 - w) * x0
        self.history.append((mask, x0, next_latents))
        return next_latents

    def step(self, step_index, latents, noise_pred, t, timesteps, target_shape, seed_g, scheduler, scheduler_kwargs, denoise):
        """`PnPHandler.step` (:276-351): returns (latents after the step | None when a model call was interrupted, scheduler)."""
        if noise_pred is None:
            return None, scheduler
        self.history, self.settled = [None], False
        sigma = t.item() / 1000.0

        def take(pred, x):                                            # scheduler step + the x0 estimate it implies
            sliced = pred[:, :x.shape[1], :target_shape[1]]
            out = scheduler.step(sliced, t, x, **scheduler_kwargs)
            nxt = out.prev_sample if hasattr(out, "prev_sample") else (out[0] if isinstance(out, (tuple, list)) else out)
            x0 = out.pred_original_sample if hasattr(out, "pred_original_sample") else x - (t.item() / 1000.0) * sliced
            return nxt, x0

        repeats = self.repeats.get(step_index, 0)
        if repeats <= 1:
            return take(noise_pred, latents)[0], scheduler
        saved = copy.deepcopy(scheduler) if (scheduler is not None and getattr(scheduler, "is_stateful", True)) else None
        result = self._absorb(latents, *take(noise_pred, latents))
        for _ in range(1, repeats):
            if self.settled:
                break
            if saved:
                scheduler = copy.deepcopy(saved)
            noise = torch.randn(latents.shape, generator=seed_g, device=latents.device, dtype=latents.dtype)
            probe = (1.0 - sigma) * self.history[-1][1] + sigma * noise
            pred = denoise(probe)
            if pred is None:
                return None, scheduler
            result = self._absorb(probe, *take(pred, probe))
        return result, scheduler


def create(plan, f_uncertainty, setting, certain_percentage, channel_dim: int = 1) -> SelfRefiner:
    """`create_self_refiner_handler(self_refiner_plan, self_refiner_f_uncertainty, self_refiner_setting,
    self_refiner_certain_percentage)` as generate() calls it (any2video.py:1486): the setting (1 / 2) doubles as the norm's p."""
    return SelfRefiner(plan, f_uncertainty, setting, certain_percentage, channel_dim)
