"""TEST INFRASTRUCTURE ONLY -- CPU statement of the adapter algebra of the checkpoint / LoRA row (SURVEY.md section 8(f)
rank 2).  PARITY UNPINNED for the `alpha / rank` factor and the multiplier m_i (the alpha-less core -- W + B A, + diff, + diff_b at
m = 1 -- IS pinned: tests/test_lora_extract_vs_golden.py merges the file the reference's own shared/extract_lora.py wrote for an
(original, finetuned) pair and must get the finetuned checkpoint back): the arithmetic lives in mmgp 3.7.12 (`requirements.txt:2`,
`offload.load_loras_into_model` / `activate_loras` and its patched `Linear.forward`), a third-party dependency that is not
part of /root/reference, so there is no reference code to execute.  Restated from its published behaviour and from the
reference's call sites (wgp.py:6893-6935; shared/utils/loras_mutipliers.py:143-148; key layout produced by
WanModel.preprocess_loras, models/wan/modules/model.py:942-1036):

    y = x W^T + b + sum_i m_i * [ (alpha_i / r_i) * (x A_i^T) B_i^T + x diff_i^T + diff_b_i ]

i.e. the layer behaves as if its weight were W + sum_i m_i ((alpha_i/r_i) B_i A_i + diff_i) and its bias b + sum_i m_i diff_b_i.
The key / multiplier functions of this row ARE pinned (tests/test_loader_host_vs_golden.py); only this algebra is not.
Only tests/ may import this module.
"""
import numpy as np
import torch


def merged_weight_exact(W, adapters, mults):
    """float64 W + sum_i m_i (scale_i B_i A_i + diff_i); adapters: list of {"A","B","alpha","diff"} (torch tensors)."""
    acc = W.double().reshape(W.shape[0], -1).clone()
    for ad, m in zip(adapters, mults):
        if m == 0:
            continue
        if "A" in ad:
            r = ad["A"].shape[0]
            s = (ad["alpha"] / r) if ad.get("alpha") is not None else 1.0
            acc += m * s * (ad["B"].double() @ ad["A"].double())
        if "diff" in ad:
            acc += m * ad["diff"].double().reshape_as(acc)
    return acc.reshape(W.shape)


def merged_bias_exact(b, adapters, mults):
    acc = b.double().clone()
    for ad, m in zip(adapters, mults):
        if m != 0 and "diff_b" in ad:
            acc += m * ad["diff_b"].double()
    return acc


def runtime_lora_linear(x, W, b, adapters, mults):
    """The run-time form (fp64): x W^T + b + sum_i m_i (s_i (x A^T) B^T + x diff^T + diff_b)."""
    x = x.double()
    y = x @ W.double().reshape(W.shape[0], -1).t()
    if b is not None:
        y = y + b.double()
    for ad, m in zip(adapters, mults):
        if m == 0:
            continue
        if "A" in ad:
            r = ad["A"].shape[0]
            s = (ad["alpha"] / r) if ad.get("alpha") is not None else 1.0
            y = y + m * s * ((x @ ad["A"].double().t()) @ ad["B"].double().t())
        if "diff" in ad:
            y = y + m * (x @ ad["diff"].double().reshape(W.shape[0], -1).t())
        if "diff_b" in ad:
            y = y + m * ad["diff_b"].double()
    return y


def dequant_i8(data, scale):
    """optimum-quanto qint8 weight: bf16(float(data) * scale[row])."""
    return (data.float() * scale.float().reshape(-1, 1)).to(torch.bfloat16)


def bf16_round_ok(got_bf16, exact64, term_mag, slack=2.0 ** -20):
    """True where `got` is a correct bf16 rounding of `exact`, allowing the fp32 accumulation an absolute error of
    `slack` x `term_mag` (the magnitude of the terms that were summed: cancellation can leave a result much smaller than
    its terms):  |got - exact| <= half a bf16 ulp of |exact| + slack * term_mag."""
    g, e = got_bf16.double(), exact64.double()
    ulp = torch.pow(2.0, torch.floor(torch.log2(e.abs().clamp_min(1e-30))) - 7)       # bf16: 8 significant bits
    return (g - e).abs() <= 0.5 * ulp * (1 + 1e-6) + slack * term_mag
