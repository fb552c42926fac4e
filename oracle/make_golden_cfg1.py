"""TEST INFRASTRUCTURE ONLY -- BASELINE.json configs[0] ("config 1": Wan2.1 t2v 1.3B, 320x512x17f,
10 steps) goldens from the REFERENCE's own WanModel + FlowUniPCMultistepScheduler.

Run in the build container (needs /root/reference; ~15 min on 8 cores):   python oracle/make_golden_cfg1.py
Writes tests/golden/cfg1_forward.npz and tests/golden/cfg1_loop.npz:

  cfg1_forward   one joint CFG forward at t = 681 of the full 1.3B model (dim 1536, 12 heads, ffn 8960, 30 layers) on the
                 latent 16x5x40x64 (L = 3,200): the reference's bf16 outputs (cond, uncond), the fp32-anchor outputs
                 (oracle, fp32 everywhere, exact softmax) and, for the error-growth table, PROBE_ROWS token rows of the
                 cond stream's hidden state after every block from both.
  cfg1_loop      the 10-step CFG UniPC loop (shift 5, guide 5 -- defaults/t2v_1.3B.json, any2video.py:1490-1733) free
                 running from the seeded noise: per-step latents (every SUB-th element) and the full final latents, from the
                 reference bf16 run and from the fp32 anchor.

Weights and inputs are not stored: oracle.wan_oracle.synth_weights / synth_inputs re-derive them from the seeds.
"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, wan_oracle as O  # noqa: E402
from oracle.make_golden import build_ref_model, ref_forward, f32  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
F_, H_, W_ = 5, 40, 64
T_FWD = 681
STEPS, SHIFT, GUIDE = 10, 5.0, 5.0
SUB = 8
BF = torch.bfloat16


def probe_rows(L, n=16):
    g = torch.Generator().manual_seed(123)
    return torch.sort(torch.randperm(L, generator=g)[:n]).values


def main():
    torch.manual_seed(0)
    ns = ref_shim.load()
    cfg = O.make_config("t2v_1.3B")
    L = F_ * (H_ // 2) * (W_ // 2)
    rows = probe_rows(L)
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, F_, H_, W_)
    t0 = time.time()
    W = O.synth_weights(cfg)
    m = build_ref_model(ns, cfg, W, BF)
    print(f"reference model built in {time.time() - t0:.1f}s", flush=True)
    which = sys.argv[1:] or ["forward", "loop"]

    if "forward" in which:
        out = {"shape": np.array([F_, H_, W_]), "t": np.array([T_FWD]), "probe_rows": rows.numpy()}
        per_layer = []
        calls = {"n": 0}

        def hook(mod, args, output):
            # the reference runs block i on stream 0 then stream 1 (model.py:1993-2036): keep stream 0 (cond)
            if calls["n"] % 2 == 0:
                per_layer.append(output[0, rows].detach().float().clone())
            calls["n"] += 1
        hs = [b.register_forward_hook(hook) for b in m.blocks]
        t = torch.tensor([T_FWD], dtype=torch.int64)
        t0 = time.time()
        r = ref_forward(ns, m, [lat, lat], t, [ctx, ctx_null])
        print(f"reference bf16 forward (2 streams): {time.time() - t0:.1f}s", flush=True)
        for h in hs:
            h.remove()
        assert len(per_layer) == cfg.num_layers, len(per_layer)
        out["cond_bf16"], out["uncond_bf16"] = f32(r[0]), f32(r[1])
        out["layers_bf16"] = torch.stack(per_layer).to(BF).view(torch.int16).numpy()      # bf16 bits [30, rows, 1536]
        # the oracle must reproduce the reference bit-for-bit here too (checked again in tests/test_cfg1_oracle_vs_golden.py)
        t0 = time.time()
        o = O.dit_forward([lat, lat], t, [ctx, ctx_null], W, cfg, dtype=BF)
        print(f"oracle bf16 forward: {time.time() - t0:.1f}s; bit-equal to the reference: "
              f"{torch.equal(o[0], r[0]) and torch.equal(o[1], r[1])}", flush=True)
        W32 = O.synth_weights(cfg, dtype=torch.float32)
        anchor_layers = []
        t0 = time.time()
        a = O.dit_forward([lat, lat], t, [ctx.float(), ctx_null.float()], W32, cfg, dtype=torch.float32, exact=True,
                          probe=lambda i, s, h: anchor_layers.append(h[0, rows].clone()) if s == 0 else None)
        print(f"fp32 anchor forward: {time.time() - t0:.1f}s", flush=True)
        out["cond_fp32"], out["uncond_fp32"] = f32(a[0]), f32(a[1])
        out["layers_fp32"] = torch.stack(anchor_layers).numpy()
        del W32
        np.savez_compressed(os.path.join(OUT, "cfg1_forward.npz"), **out)
        for k in ("cond", "uncond"):
            e = np.linalg.norm(out[k + "_bf16"] - out[k + "_fp32"]) / np.linalg.norm(out[k + "_fp32"])
            print(f"  {k}: |ref_bf16 - fp32| / |fp32| = {e:.4e}")
        print("cfg1_forward.npz", {k: v.shape for k, v in out.items()}, flush=True)

    if "loop" in which:
        out = {"shape": np.array([F_, H_, W_]), "steps": np.array([STEPS]), "shift": np.array([SHIFT]),
               "guide": np.array([GUIDE]), "sub": np.array([SUB])}
        s = ns.U.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(STEPS, device="cpu", shift=SHIFT)
        latents = lat.clone()
        sub = []
        t0 = time.time()
        for i, t in enumerate(s.timesteps):
            cond, uncond = ref_forward(ns, m, [latents, latents], torch.stack([t]), [ctx, ctx_null])
            noise = uncond + GUIDE * (cond - uncond)                       # any2video.py:1722
            latents = s.step(noise, t, latents, return_dict=False)[0]     # :1733
            sub.append(f32(latents).reshape(-1)[::SUB].copy())
            print(f"  reference step {i} t={int(t)} |x|={latents.norm():.3f} ({time.time() - t0:.0f}s)", flush=True)
        out["timesteps"] = s.timesteps.numpy().copy()
        out["sub_bf16"] = np.stack(sub)
        out["final_bf16"] = f32(latents)
        del m
        W32 = O.synth_weights(cfg, dtype=torch.float32)
        sub32 = []
        t0 = time.time()
        fin, _ = O.sample_loop(W32, cfg, lat.clone(), ctx.float(), ctx_null.float(), STEPS, SHIFT, GUIDE, dtype=torch.float32,
                               exact=True, on_step=lambda i, x: (sub32.append(f32(x).reshape(-1)[::SUB].copy()),
                                                                 print(f"  anchor step {i} ({time.time() - t0:.0f}s)", flush=True)))
        out["sub_fp32"] = np.stack(sub32)
        out["final_fp32"] = f32(fin)
        np.savez_compressed(os.path.join(OUT, "cfg1_loop.npz"), **out)
        for i in range(STEPS):
            e = np.linalg.norm(out["sub_bf16"][i] - out["sub_fp32"][i]) / np.linalg.norm(out["sub_fp32"][i])
            print(f"  step {i}: |ref_bf16 - fp32| / |fp32| = {e:.4e}")
        print("cfg1_loop.npz", {k: v.shape for k, v in out.items()}, flush=True)


if __name__ == "__main__":
    main()
