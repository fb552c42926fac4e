"""-m gpu: CFG parallelism (wan2gp_amd/sp.py `CfgParallel`) end to end on the HIP path, all ranks on cuda:0 with a `gloo` process
group staging the exchanges through the host (the test boxes have one GPU; on a multi-GPU node the same code runs over RCCL):
world 2 = the conditional and the unconditional stream in two processes, no sequence parallelism; world 4 = two halves of two
sequence-parallel ranks each -- the single-stream forward with the half's K / V^T gathers; the same world with the Ulysses exchange,
and world 8 = two halves of four ranks with the Ulysses exchange (`cfg2 x sp4 (ulysses)`: what `bench.py --gpus 8` runs by default) --
the ONE-stream form of the all-to-all path (no V^T block swap, world x 1 query batches), one head per rank.  Every rank must end up
with the (cond, uncond) pair of the single-rank joint pass.  (Named to run after the other -m gpu files.)"""
import datetime
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, mode="allgather"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    # a bounded collective timeout: a rank that fails before a swap must not leave its partner (and the test run) waiting for the
    # backend's 30-minute default
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        from oracle import wan_oracle as O
        from wan2gp_amd.model import WanModelHIP
        from wan2gp_amd.sp import CfgParallel
        cfg = O.make_config("small")
        W = O.synth_weights(cfg, seed=77)
        m = WanModelHIP(dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers)
        m.load_state_dict(W)
        f, h, w = 4, 12, 16                    # L = 192 tokens; world 4: 96 per rank of a half (not a multiple of 64)
        lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w, seed=9)
        t = torch.tensor([412])
        ref = m([lat.cuda(), lat.cuda()], t=t, context=[ctx.cuda(), ctx_null.cuda()])        # the joint pass on one rank
        cfgp = CfgParallel(rank, world, mode=mode).attach(m)
        assert (m.sp is None) == (world == 2) and (m.sp is None or (m.sp.mode == mode and m.sp.world == world // 2))
        got = cfgp.guided_pair(m, lat.cuda(), ctx.cuda(), ctx_null.cuda(), t=t)
        for name, g, r in zip(("cond", "uncond"), got, ref):
            assert g.shape == r.shape and g.is_cuda
            rel = ((g - r).norm() / r.norm()).item()
            assert rel < 1e-2, f"rank {rank} (stream {cfgp.stream}): {name} deviates from the joint pass: rel={rel}"
        # both ranks of a pair hold the SAME bits (the combine and the scheduler then keep the latents replicated)
        mine = torch.stack([g.cpu() for g in got])
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        assert all(torch.equal(p, parts[0]) for p in parts), "the ranks do not hold identical (cond, uncond) pairs"
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "allgather"), (4, "allgather"), (4, "ulysses"), (8, "ulysses")])
def test_cfg_parallel_pair_matches_the_joint_pass(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=900) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:                        # a worker stuck in a collective must not outlive the test (nor block interpreter exit)
            if p.is_alive():
                p.kill()
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
