#!/usr/bin/env python
"""One CFG step of the headline workload on one GPU, then the SAME step as rank 0 of a simulated world of N (bench.py's
simulate_world: the exchanges are device-to-device copies on a side stream), with marker launches between the phases so that a
`rocprofv3 --kernel-trace` of this process can be cut into "full step" and "rank step" and compared kernel class by kernel class
(tools/rank_trace_table.py).  Round 6, VERDICT item 1a: where do the 116 ms over step / 8 go?

usage (under rocprofv3): python tools/rank_trace.py [--world 8] [--layout ulysses] [--workload 14B-720p] [--chunks 0] [--steps 1]
Markers: `delay_kernel` launches (wan_debug_delay 1 us) -- 3 in a row open the full-step phase, 5 the rank phase, 7 close the trace."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--layout", default="ulysses")
    ap.add_argument("--workload", default="14B-720p")
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--fp8", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    from wan2gp_amd import lib as L_
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.rope import get_rotary_pos_embed
    from wan2gp_amd.schedulers import HipScheduler, cfg_combine
    cfg, (f, h, w), desc = bench.WORKLOADS[args.workload]
    L = f * (h // 2) * (w // 2)
    model = bench.random_weights(WanModelHIP(**cfg), cfg, 1234, args.fp8)
    lib = L_.load()
    g = torch.Generator(device="cuda").manual_seed(42)
    ctx = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx[:, 77:] = 0
    ctx_null = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx_null[:, 8:] = 0
    freqs = get_rotary_pos_embed((f, h, w), device="cuda")
    y = None
    if cfg.get("in_dim", 16) == 36:
        y = torch.randn(20, f, h, w, device="cuda", generator=g)
    latents = torch.randn(1, 16, f, h, w, device="cuda", generator=g)
    par = {"cfgp": None}

    def new_sched(n_steps=bench.VIDEO_STEPS):
        sc = HipScheduler("unipc", num_train_timesteps=1000)
        sc.set_timesteps(n_steps, device="cuda", shift=12.0)
        return sc

    def one_step(i, lat, sc=None, fr=None):
        t = sc.timesteps[i]
        fr = freqs if fr is None else fr
        if par["cfgp"] is not None:
            cond, uncond = par["cfgp"].guided_pair(model, lat, ctx, ctx_null, t=torch.stack([t]), freqs=fr, y=y)
        else:
            cond, uncond = model([lat, lat], t=torch.stack([t]), context=[ctx, ctx_null], freqs=fr, y=y)
        return sc.step(cfg_combine(cond, uncond, 4.0), t, lat)[0]

    def marker(n):
        torch.cuda.synchronize()
        for _ in range(n):
            L_.check(lib.wan_debug_delay(1.0, L_.stream_ptr()), "wan_debug_delay")
        torch.cuda.synchronize()

    sc = new_sched()
    lat = one_step(0, latents, sc)
    marker(3)
    t0 = time.perf_counter()
    for i in range(args.steps):
        lat = one_step(1 + i, lat, sc)
    torch.cuda.synchronize()
    step_s = (time.perf_counter() - t0) / args.steps
    marker(5)
    res = bench.simulate_world([args.world], model, None, one_step, latents, new_sched, step_s, cfg, L, par, args.layout, None, args.steps, 0.0,
                               args.chunks or None)
    marker(7)
    print(json.dumps({"workload": desc, "one_gpu_step_ms": step_s * 1e3, "steps_per_phase": args.steps,
                      "rank_phase_steps": args.steps + 1, "simulated": res}), flush=True)


if __name__ == "__main__":
    main()
