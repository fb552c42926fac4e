"""LoRA files -> merged resident weights.

Reference path: `wgp.py:6893-6935` parses the multiplier string (`shared/utils/loras_mutipliers.py:3-140`), normalises
the LoRA state dict (`WanModel.preprocess_loras`, `models/wan/modules/model.py:942-1036`) and hands it to
`mmgp.offload.load_loras_into_model`; per denoising step `update_loras_slists` -> `offload.activate_loras`
(`loras_mutipliers.py:143-148`) selects the step's multipliers and mmgp's patched `Linear.forward` adds
`m * (alpha / rank) * (x A^T) B^T` (+ `m * diff`, `m * diff_b`) to every adapted layer on every call.

MI355X design: nothing is offloaded (288 GB HBM), so the adapters are **merged** into the resident bf16 weights once,
`W <- bf16(W + sum_i m_i * (alpha_i / r_i) * B_i A_i)`, by `wan_lora_merge` (fp32 accumulate, one rounding), and the hot
path runs unchanged at full speed -- no skinny GEMM pair per Linear per step.  When a step's multipliers differ from the
merged ones (phase switch inside one expert, per-step lists) `set_step` merges the *difference* (the base weights are
kept to re-merge from scratch, so no rounding drift accumulates).  The two Wan 2.2 experts have their own weights, which
covers the Lightning "1;0 0;1" profiles (`profiles/wan_2_2/*.json`) without any re-merge.

mmgp 3.7.12 (requirements.txt:2) is not part of /root/reference: the adapter algebra above is its published behaviour
restated; parity for the `alpha / rank` factor and the multiplier is therefore against the oracle only ("parity unpinned", DESIGN.md
section 4) -- the alpha-less core (W + B A, + diff, + diff_b) is pinned to the reference's own adapter-file producer, shared/extract_lora.py:13-30,
by an extract -> merge round trip (tests/golden/lora_extract.npz) --
while every key / multiplier function below is pinned to the reference's own code through tests/golden/loader_golden.json.
"""
from typing import Dict, List, Optional

import torch

from .lib import WanHipError

# ---------------------------------------------------------------------------------------------------------------------
# multiplier strings (loras_mutipliers.py)
# ---------------------------------------------------------------------------------------------------------------------


def preparse_loras_multipliers(text):
    """loras_mutipliers.py:4-12: drop comment lines, join lines, `|` is a blank; -> list of per-LoRA strings."""
    if isinstance(text, list):
        return [m.strip(" \r\n") if isinstance(m, str) else m for m in text]
    lines = [ln.strip() for ln in text.strip(" \r\n").replace("\r", "").split("\n")]
    joined = " ".join(ln for ln in lines if ln and not ln.startswith("#"))
    return joined.replace("|", " ").strip().split(" ")


def _spread(values, n):
    """n samples of `values` at stride len/n (loras_mutipliers.py:15-26)."""
    if not isinstance(values, list):
        values = [values]
    if n <= 0:
        return []
    stride = len(values) / n
    out, pos = [], 0
    for _ in range(n):
        out.append(values[int(pos)])
        pos += stride
    return out


def expand_slist(slists, lora_no, num_inference_steps, model_switch_step, model_switch_step2):
    """Per-step multipliers of one LoRA: a float when constant, else a list of `num_inference_steps` values
    (loras_mutipliers.py:14-37)."""
    p1, p2, p3 = (slists[f"phase{i}"][lora_no] for i in (1, 2, 3))
    if slists["shared"][lora_no]:
        return p1 if isinstance(p1, float) else _spread(p1, num_inference_steps)
    if all(isinstance(p, float) for p in (p1, p2, p3)) and p1 == p2 == p3:
        return p1
    return (_spread(p1, model_switch_step) + _spread(p2, model_switch_step2 - model_switch_step)
            + _spread(p3, num_inference_steps - model_switch_step2))


def _blank_slists(n):
    return {"phase1": [1.] * n, "phase2": [1.] * n, "phase3": [1.] * n, "shared": [False] * n}


def _as_float(s):
    try:
        return float(s)
    except (TypeError, ValueError):
        return None


def parse_loras_multipliers(loras_multipliers, nb_loras, num_inference_steps, merge_slist=None, nb_phases=2,
                            model_switch_step=None, model_switch_step2=None, model_switch_phase=1,
                            lora_multiplier_branches=None):
    """loras_mutipliers.py:48-140.  Returns (first-step multipliers, slists dict, error string) -- ("", "", msg) on a
    malformed string, exactly as the reference does."""
    if isinstance(loras_multipliers, str) and loras_multipliers.count("|") > 1:
        return "", "", "There can be only one '|' character in Loras Multipliers Sequence"
    if model_switch_step is None:
        model_switch_step = num_inference_steps
    if model_switch_step2 is None:
        model_switch_step2 = num_inference_steps
    branches = [str(b).strip() for b in (lora_multiplier_branches or []) if str(b).strip()]
    slists = {"model_switch_step": model_switch_step, "model_switch_step2": model_switch_step2, **_blank_slists(nb_loras)}
    for b in branches:
        slists[b] = _blank_slists(nb_loras)
    targets = [slists[b] for b in branches] if branches else [slists]

    def parse_one(mult, lora_no, phase_no):
        if "," in mult:
            pieces = mult.split(",")
            vals = []
            for piece in pieces:
                f = _as_float(piece)
                if f is None:
                    return None, (f"Lora sub value no {lora_no + 1} ({piece}) in Multiplier definition '{pieces}' is invalid in "
                                  f"Phase {phase_no + 1}")
                vals.append(f)
            return vals, ""
        f = _as_float(mult)
        if f is None:
            return None, f"Lora Multiplier no {lora_no + 1} ({mult}) is invalid"
        return f, ""

    def store(tgt, lora_no, phase_no, value, shared):
        if shared:
            tgt["phase1"][lora_no] = tgt["phase2"][lora_no] = tgt["phase3"][lora_no] = value
            tgt["shared"][lora_no] = True
        else:
            tgt[f"phase{phase_no + 1}"][lora_no] = value

    if isinstance(loras_multipliers, list) or len(loras_multipliers) > 0:
        for i, mult in enumerate(preparse_loras_multipliers(loras_multipliers)[:nb_loras]):
            if not isinstance(mult, str):
                for tgt in targets:
                    store(tgt, i, 0, float(mult), True)
                continue
            phases = mult.strip().split(";")
            shared = len(phases) <= 1
            if not shared and len(phases) != nb_phases:
                if len(phases) > nb_phases:
                    return "", "", (f"if the ';' syntax is used for one Lora multiplier, there should be at most {nb_phases} phases "
                                    f"for this multiplier")
                phases = (phases[:1] + phases) if model_switch_phase == 2 else (phases + phases[-1:])
            for phase_no, pm in enumerate(phases):
                if branches and ":" in pm:
                    per_branch = pm.split(":")
                    if len(per_branch) != len(branches):
                        return "", "", f"Lora Multiplier no {i + 1} ({pm}) should define {len(branches)} branch values separated by ':'"
                    for b, bm in zip(branches, per_branch):
                        val, err = parse_one(bm, i, phase_no)
                        if err:
                            return "", "", err
                        store(slists[b], i, phase_no, val, shared)
                else:
                    val, err = parse_one(pm, i, phase_no)
                    if err:
                        return "", "", err
                    for tgt in targets:
                        store(tgt, i, phase_no, val, shared)

    keys = ("phase1", "phase2", "phase3", "shared")
    if merge_slist is not None:
        for tgt, src in ([(slists[b], merge_slist[b]) for b in branches] if branches else [(slists, merge_slist)]):
            for k in keys:
                tgt[k] = src[k] + tgt[k]
    if branches:
        for k in keys:
            slists[k] = slists[branches[0]][k]
    first = []
    for i in range(len(slists["phase1"])):
        e = expand_slist(slists, i, num_inference_steps, model_switch_step, model_switch_step2)
        first.append(e[0] if isinstance(e, list) else e)
    return first, slists, ""


def get_model_switch_steps(timesteps, guide_phases, model_switch_phase, switch_threshold, switch2_threshold):
    """First step index whose timestep is <= the threshold, per phase boundary (loras_mutipliers.py:152-168)."""
    n = len(timesteps)
    s1 = s2 = None
    for i, t in enumerate(timesteps):
        if guide_phases >= 2 and s1 is None and t <= switch_threshold:
            s1 = i
        if guide_phases >= 3 and s2 is None and t <= switch2_threshold:
            s2 = i
    s1 = n if s1 is None else s1
    s2 = n if s2 is None else s2
    desc = ""
    if guide_phases > 1:
        desc = "Denoising Steps: " + (" Phase 1 = None" if s1 == 0 else f" Phase 1 = 1:{min(s1, n)}")
        if s1 < n:
            desc += ", Phase 2 = None" if s1 == s2 else f", Phase 2 = {s1 + 1}:{min(s2, n)}"
            if guide_phases > 2 and s2 < n:
                desc += f", Phase 3 = {s2 + 1}:{n}"
    return s1, s2, desc


def step_multipliers(slists, num_inference_steps, step_no, model_switch_step=None, model_switch_step2=None):
    """What `update_loras_slists` + `offload.activate_loras` make active at `step_no` (loras_mutipliers.py:143-148)."""
    s1 = slists["model_switch_step"] if model_switch_step is None else model_switch_step
    s2 = slists["model_switch_step2"] if model_switch_step2 is None else model_switch_step2
    out = []
    for i in range(len(slists["phase1"])):
        e = expand_slist(slists, i, num_inference_steps, s1, s2)
        out.append(e[step_no] if isinstance(e, list) else e)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# LoRA state-dict normalisation (WanModel.preprocess_loras)
# ---------------------------------------------------------------------------------------------------------------------
_KOHYA_TOP = [(src, "diffusion_model." + dst) for name, dst in (
    ("head_head", "head.head"), ("img_emb_proj_", "img_emb.proj."), ("text_embedding_", "text_embedding."),
    ("time_embedding_", "time_embedding."), ("time_projection_", "time_projection."))
    for src in ("lora_unet__" + name, "lora_unet_" + name)]
_I2V_ONLY = ("cross_attn.k_img", "cross_attn.v_img", "img_emb.")


def normalize_lora_keys(sd: Dict[str, torch.Tensor], base_model_type="t2v", i2v_class=False, vace_layers=None):
    """`WanModel.preprocess_loras(base_model_type, sd)` (model.py:942-1036).  `i2v_class` is what the reference looks up
    with `wgp.test_class_i2v(base_model_type)`; `vace_layers` the model's VACE block map (model.py:1003-1013)."""
    first = next(iter(sd), None)
    if first is None:
        return sd
    if base_model_type == "scail":
        sd.pop("diffusion_model.patch_embedding.diff", None)
        sd.pop("diffusion_model.patch_embedding.diff_b", None)
        return sd
    sd = {k: v for k, v in sd.items() if not k.endswith("modulation.diff")}
    if ".default." in first:
        sd = {k.replace(".default.", "."): v for k, v in sd.items()}
    if first.startswith("vace_blocks."):
        out = {}
        for k, v in sd.items():
            if k.startswith("vace_blocks."):
                parts = k.split(".")
                parts[0], parts[1] = "blocks." + str(vace_layers[int(parts[1])]), "vace"
                k = ".".join(parts)
            out[k] = v
        sd = out
    if first.startswith("lora_unet_"):                       # kohya naming -> diffusers naming
        out = {}
        for k, v in sd.items():
            k = k.replace("lora_unet_blocks_", "diffusion_model.blocks.").replace("lora_unet__blocks_", "diffusion_model.blocks.")
            for src, dst in _KOHYA_TOP:
                k = k.replace(src, dst)
            for part in ("cross_attn", "self_attn", "ffn"):
                k = k.replace("_" + part + "_", "." + part + ".")
            out[k.replace("lora_up", "lora_B").replace("lora_down", "lora_A")] = v
        sd = out
    if base_model_type in ("scail2_14B", "scail2_1.3B"):
        sd = {k: v for k, v in sd.items()
              if not ("patch_embedding.diff" in k and torch.is_tensor(v) and v.ndim >= 2 and v.shape[1] != 20)}
    if not i2v_class or base_model_type == "i2v_2_2":
        sd = {k: v for k, v in sd.items() if not any(tag in k for tag in _I2V_ONLY)}
    return sd


# ---------------------------------------------------------------------------------------------------------------------
# adapters -> modules, merge
# ---------------------------------------------------------------------------------------------------------------------
_PREFIXES = ("diffusion_model.", "transformer.", "model.diffusion_model.")
_SUFFIXES = ((".lora_A.weight", "A"), (".lora_B.weight", "B"), (".lora_down.weight", "A"), (".lora_up.weight", "B"),
             (".alpha", "alpha"), (".diff_b", "diff_b"), (".diff", "diff"))


def group_adapter(sd: Dict[str, torch.Tensor], errors: Optional[List[str]] = None):
    """{module name: {"A": [r,K], "B": [N,r], "alpha": float | None, "diff": [N,K] | None, "diff_b": [N] | None}} from
    normalised keys `[diffusion_model.]<module>.lora_A.weight` ... (the layout load_loras_into_model consumes).
    With `errors` (a list) problems are appended there and the offending key / module is skipped -- the reference
    collects per-key problems in `_loras_errors` and keeps loading; without it they raise."""
    mods: Dict[str, dict] = {}
    for k, v in sd.items():
        for p in _PREFIXES:
            if k.startswith(p):
                k = k[len(p):]
                break
        for suf, slot in _SUFFIXES:
            if k.endswith(suf):
                mods.setdefault(k[: -len(suf)], {})[slot] = float(v) if slot == "alpha" else v
                break
        else:
            msg = f"LoRA key {k!r}: unknown suffix (expected lora_A/lora_B/lora_down/lora_up/alpha/diff/diff_b)"
            if errors is None:
                raise WanHipError(msg)
            errors.append(msg)
    for name in list(mods):
        m = mods[name]
        msg = None
        if ("A" in m) != ("B" in m):
            msg = f"LoRA module {name!r}: lora_A without lora_B (or the reverse)"
        elif "A" in m and m["A"].shape[0] != m["B"].shape[1]:
            msg = f"LoRA module {name!r}: rank mismatch A {list(m['A'].shape)} / B {list(m['B'].shape)}"
        if msg is not None:
            if errors is None:
                raise WanHipError(msg)
            errors.append(msg)
            del mods[name]
    return mods


def adapter_scale(m):
    """alpha / rank, 1 when the file carries no alpha."""
    if "A" not in m:
        return 1.0
    a = m.get("alpha")
    return 1.0 if a is None else a / m["A"].shape[0]


class MergedLoras:
    """Adapters of one resident model (`WanModelHIP` or anything exposing `_weights: {key: bf16 HBM tensor}`)."""

    def __init__(self, model):
        self.model = model
        self.adapters: List[Dict[str, dict]] = []
        self.merged: List[float] = []
        self._base: Dict[str, torch.Tensor] = {}          # pristine copies of every weight an adapter touches
        self.errors: List[str] = []

    def _target(self, module, what):
        w = self.model._weights.get(f"{module}.{what}")
        if w is None:
            self.errors.append(f"{module}.{what}: no such tensor in the model")
        return w

    def add(self, sd, base_model_type="t2v", i2v_class=False, normalized=False):
        """One LoRA file (state dict).  Returns its index; multipliers are applied by `set_multipliers`."""
        if not normalized:
            sd = normalize_lora_keys(dict(sd), base_model_type, i2v_class, getattr(self.model, "vace_layers", None))
        mods = group_adapter(sd, self.errors)
        dev = self.model.device
        for name, m in mods.items():
            for slot in ("A", "B", "diff", "diff_b"):
                if slot in m:
                    m[slot] = m[slot].to(device=dev, dtype=torch.float32).contiguous()
            w = self._target(name, "weight") if ("A" in m or "diff" in m) else None
            if w is None:
                m.pop("A", None); m.pop("B", None); m.pop("diff", None)
            else:
                want = (m["B"].shape[0], m["A"].shape[1]) if "A" in m else tuple(m["diff"].shape)
                if tuple(w.shape[:2]) != want and w.numel() != want[0] * want[1]:
                    self.errors.append(f"{name}: adapter shape {want} does not match weight {list(w.shape)}")
                    m.pop("A", None); m.pop("B", None); m.pop("diff", None)
                elif w.dtype != torch.bfloat16:
                    self.errors.append(f"{name}: fp32-locked tensor (patch_embedding / head) -- adapter skipped")
                    m.pop("A", None); m.pop("B", None); m.pop("diff", None)
            if "diff_b" in m:
                b = self._target(name, "bias")
                if b is None or b.dtype != torch.bfloat16:
                    m.pop("diff_b")
        self.adapters.append(mods)
        self.merged.append(0.0)
        return len(self.adapters) - 1

    def set_multipliers(self, mults):
        """Make the resident weights equal bf16(base + sum_i mults[i] * delta_i).  No-op when nothing changed."""
        from . import ops
        mults = [float(m) for m in mults] + [0.0] * (len(self.adapters) - len(mults))
        if mults[: len(self.adapters)] == self.merged:
            return
        touched = {}
        for mods in self.adapters:
            for name, m in mods.items():
                if "A" in m or "diff" in m:
                    touched[f"{name}.weight"] = True
                if "diff_b" in m:
                    touched[f"{name}.bias"] = True
        for key in touched:
            w = self.model._weights[key]
            if key not in self._base:
                self._base[key] = w.clone()
            else:
                w.copy_(self._base[key])
        # fp32 sum of all deltas per tensor, then a single rounding: accumulate in an fp32 scratch per tensor
        acc: Dict[str, torch.Tensor] = {}
        for mods, mult in zip(self.adapters, mults):
            if mult == 0.0:
                continue
            for name, m in mods.items():
                if "A" in m or "diff" in m:
                    key = f"{name}.weight"
                    w = self.model._weights[key]
                    a = acc.get(key)
                    if a is None:
                        a = acc[key] = torch.zeros(w.shape[0], w.numel() // w.shape[0], dtype=torch.float32, device=w.device)
                    if "A" in m:
                        ops.lora_accumulate(a, m["B"], m["A"], mult * adapter_scale(m))
                    if "diff" in m:
                        ops.axpy_f32(a, m["diff"].view_as(a), mult)
                if "diff_b" in m:
                    key = f"{name}.bias"
                    a = acc.get(key)
                    if a is None:
                        a = acc[key] = torch.zeros(1, self.model._weights[key].numel(), dtype=torch.float32, device=self.model.device)
                    ops.axpy_f32(a, m["diff_b"].view_as(a), mult)
        for key, a in acc.items():
            ops.add_f32_into_bf16_(self.model._weights[key], a)
        self.merged = mults[: len(self.adapters)]

    def set_step(self, slists, num_inference_steps, step_no, model_switch_step=None, model_switch_step2=None):
        """`update_loras_slists` for a merged model: re-merge only if this step's multipliers differ."""
        self.set_multipliers(step_multipliers(slists, num_inference_steps, step_no, model_switch_step, model_switch_step2))

    def unload(self):
        """`offload.unload_loras_from_model`: restore the pristine weights."""
        for key, base in self._base.items():
            self.model._weights[key].copy_(base)
        self._base.clear(); self.adapters.clear(); self.merged = []


def load_loras_into_model(model, loras, base_model_type="t2v", i2v_class=False, multipliers=None):
    """`offload.load_loras_into_model(trans, files, mults, preprocess_sd=...)` for a resident model (wgp.py:6922-6931):
    `loras` are safetensors paths or state dicts; attaches `model.loras` (a `MergedLoras`) and merges the first-step
    multipliers if given.  Errors (unknown modules, shape mismatches) are collected in `model.loras.errors` like the
    reference's `trans._loras_errors`."""
    from .checkpoint import read_safetensors
    ml = getattr(model, "loras", None) or MergedLoras(model)
    for item in loras:
        ml.add(read_safetensors(item) if isinstance(item, (str, bytes)) else item, base_model_type, i2v_class)
    model.loras = ml
    if multipliers is not None:
        ml.set_multipliers(multipliers)
    return ml
