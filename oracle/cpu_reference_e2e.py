#!/usr/bin/env python
"""TEST / BASELINE INFRASTRUCTURE ONLY (lives under oracle/ like the golden generators: nothing in the product imports it).
BASELINE.md section 3 as written: BASELINE configs[0] (Wan2.1 t2v 1.3B, 320x512x17f, 10 steps, unipc, shift 5, guidance 5)
END TO END on host cores through the REFERENCE'S OWN modules -- `WanModel`, `FlowUniPCMultistepScheduler`, `WanVAE_` executed from
/root/reference through oracle/ref_shim.py -- noise -> 10 CFG steps (20 forwards) -> VAE decode -> uint8 [3,17,320,512], every
step timed, nothing composed or extrapolated.

    python oracle/cpu_reference_e2e.py [out.json]        (build container only: needs /root/reference; ~10 minutes on 8 cores)

`bench.py`'s `cpu_baseline` leg cannot do this -- the reference tree does not travel to the GPU box and the leg is bounded to one step
+ one decode of the bit-exact port -- so this measurement lives under profiles/ as its own file, from the machine it ran on (named in
the JSON: NOT the GPU box's host).  Baseline only: not a target, not part of the product, imports oracle/ as the checker's shim.
Synthetic weights / inputs as in SURVEY.md section 8(d) (oracle.wan_oracle.synth_weights / synth_inputs, the seeds the goldens use)."""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, vae_oracle as VO, wan_oracle as O  # noqa: E402
from oracle.make_golden import build_ref_model, ref_forward  # noqa: E402
from oracle.make_golden_vae import build_ref_vae  # noqa: E402

F_, H_, W_ = 5, 40, 64
STEPS, SHIFT, GUIDE = 10, 5.0, 5.0


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def main(out_path):
    assert ref_shim.available(), "needs the reference tree"
    ns = ref_shim.load()
    cfg = O.make_config("t2v_1.3B")
    t0 = time.perf_counter()
    W = O.synth_weights(cfg)
    m = build_ref_model(ns, cfg, W, torch.bfloat16)
    vae = build_ref_vae(ns, VO.synth_vae_weights())
    build_s = time.perf_counter() - t0
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, F_, H_, W_)
    scale = VO.default_scale()
    with torch.no_grad():
        ref_forward(ns, m, [lat[:, :, :1, :8, :8].clone()], torch.tensor([500]), [ctx])          # warm-up forward, untimed
        t_video = time.perf_counter()
        s = ns.U.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(STEPS, device="cpu", shift=SHIFT)
        latents = lat.clone()
        step_s = []
        for i, t in enumerate(s.timesteps):
            t0 = time.perf_counter()
            cond, uncond = ref_forward(ns, m, [latents, latents], torch.stack([t]), [ctx, ctx_null])   # the joint pass, any2video.py:1626-1634
            noise = uncond + GUIDE * (cond - uncond)                                                  # :1722
            latents = s.step(noise, t, latents, return_dict=False)[0]                                 # :1733
            step_s.append(time.perf_counter() - t0)
            print(f"step {i}: {step_s[-1]:.1f} s", flush=True)
        t0 = time.perf_counter()
        video = vae.decode_to_cpu_uint8(latents.float(), scale, 0)                                    # vae.py:741-767 -> uint8
        dec_s = time.perf_counter() - t0
        total = time.perf_counter() - t_video
    video = video[0] if video.dim() == 5 else video
    assert video.dtype == torch.uint8 and tuple(video.shape) == (3, (F_ - 1) * 4 + 1, H_ * 8, W_ * 8), (video.dtype, video.shape)
    L = F_ * (H_ // 2) * (W_ // 2)
    rec = {"what": "BASELINE configs[0] end to end on host cores through the reference's own WanModel / FlowUniPCMultistepScheduler / WanVAE_ "
                   "(BASELINE.md section 3): noise -> 10 CFG steps (20 forwards) -> VAE decode -> uint8, every step timed",
           "kind": "reference", "workload": f"Wan2.1 t2v 1.3B 320x512x17f (latent 16x{F_}x{H_}x{W_}, L={L}), unipc, shift {SHIFT}, guidance {GUIDE}, bf16 weights",
           "machine": "build container (NOT the GPU box's host)", "cpu_model": cpu_model(), "cores": torch.get_num_threads(),
           "os_cpu_count": os.cpu_count(), "torch": torch.__version__,
           "e2e_s_per_video": total, "denoise_steps_per_s": STEPS / sum(step_s), "forwards_per_s": 2 * STEPS / sum(step_s),
           "step_s": step_s, "median_step_s_2_to_N": statistics.median(step_s[1:]), "vae_decode_to_uint8_s": dec_s, "model_build_s_untimed": build_s,
           "video": list(video.shape), "data": "synthetic (seeded weights and inputs, SURVEY.md section 8d)"}
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "step_s"}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_cpu_reference_e2e_configs0.json"))
