"""-m gpu: skip-layer guidance on the HIP forward (`WanModelHIP.forward(perturbation_layers=...)`; any2video.py:1502,
model.py:2025-2028) against tests/golden/nag.npz: slg_* -- the reference's own WanModel run with perturbation_layers=[1]
(oracle/make_golden_nag.py).  The listed block runs for the conditional stream only; on the device that is the single-stream run
of the block chain that step skipping uses (tests/test_gpu_skipcache.py), composed by host logic that
tests/test_dit_host_logic_cpu.py checks launch by launch.  Tolerance: the forward criterion of tests/test_gpu_model.py."""
import os

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "nag.npz")))


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def test_forward_with_perturbation_layers_vs_reference_golden():
    from wan2gp_amd.model import WanModelHIP
    cfg = O.make_config("small")
    W = O.synth_weights(cfg)
    m = WanModelHIP(dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers).load_state_dict(W)
    lat, c, cn, _ = O.synth_inputs(cfg, 3, 10, 14)
    t = torch.tensor([412])
    outs = m([lat.cuda(), lat.cuda()], t=t, context=[c.cuda(), cn.cuda()], perturbation_layers=[1])
    W32 = O.synth_weights(cfg, dtype=torch.float32)
    anchor = O.dit_forward([lat, lat], t, [c.float(), cn.float()], W32, cfg, dtype=torch.float32, exact=True, perturbation_layers=[1])
    for o, key, a in zip(outs, ("slg_small_cond", "slg_small_uncond"), anchor):
        ref = torch.from_numpy(G[key])
        err_ref, err_hip = rel(ref, a), rel(o.cpu(), a)
        print(f"slg {key}: err_ref={err_ref:.4e} err_hip={err_hip:.4e} hip-vs-ref={rel(o.cpu(), ref):.4e}")
        assert err_hip <= 1.5 * err_ref + 2e-3 and rel(o.cpu(), ref) <= 2.5e-2
    plain = m([lat.cuda(), lat.cuda()], t=t, context=[c.cuda(), cn.cuda()])
    # the conditional stream is untouched by the guidance (a block run on one stream instead of two may pick another GEMM tile
    # shape: equal up to accumulation order), the unconditional one skipped a block
    assert rel(plain[0].cpu(), outs[0].cpu()) <= 1e-2
    assert rel(plain[1].cpu(), outs[1].cpu()) > 4 * rel(plain[0].cpu(), outs[0].cpu()) + 1e-2
    # the unconditional call of a non-joint pass (x_id 1) skips the listed block as well: same result as the joint pass
    solo = m([lat.cuda()], t=t, context=[cn.cuda()], perturbation_layers=[1], x_id=1)[0]
    assert rel(solo.cpu(), outs[1].cpu()) <= 1e-2
