"""CPU, gloo, world 2 and 4: CFG parallelism (wan2gp_amd/sp.py `CfgParallel`) -- the conditional and the unconditional stream of a
guided step on the two halves of the world, sequence-parallel subgroups inside each half, one 2-rank swap per step.  Checked:
the group layout (who runs which stream, who shares a sequence-parallel group, who swaps with whom), that the swap hands every
rank the (cond, uncond) pair in stream order, and that `WanAny2VHIP.generate` driven this way returns on EVERY rank exactly the
latents of the single-process joint pass (a deterministic stand-in for the DiT; no arithmetic claim about the kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class StreamDiT:
    """Stands in for WanModelHIP: one prediction per stream that depends on the stream's latent, its context and t only -- so the
    joint pass and the two single passes of different processes must agree bit for bit."""

    def __init__(self):
        self.out_dim, self.cache, self.loras, self.sp, self.calls = 16, None, None, None, []
        self.device = torch.device("cpu")

    def __call__(self, x, t, context, **kw):
        xs = list(x)
        x.clear()
        self.calls.append((len(xs), kw.get("x_id", 0), float(context[0].float().mean())))
        f = torch.cos(t.flatten()[0].float() / 1000.0)
        return [0.1 * u.float() * f + c.float().mean() + 0.01 for u, c in zip(xs, context)]


def _stub_ops():
    from wan2gp_amd import ops

    def lincomb(tensors, coefs, out=None):
        r = sum(float(c) * t_.float() for c, t_ in zip(coefs, tensors))
        return r if out is None else out.copy_(r)
    ops.lincomb = lincomb
    ops.cfg_combine = lambda c, u, g, out=None: u + g * (c - u)


def _generate(pipe, **kw):
    ctx = torch.full((1, 512, 4096), 0.5, dtype=torch.bfloat16)
    ctx_null = torch.zeros(1, 512, 4096, dtype=torch.bfloat16)
    return pipe.generate(context=ctx, context_null=ctx_null, width=64, height=64, frame_num=9, sampling_steps=5, guide_scale=4.0,
                         seed=11, return_latents=True, sample_solver="unipc", **kw)["latents"]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from wan2gp_amd.pipeline import WanAny2VHIP
        from wan2gp_amd.sp import CfgParallel
        _stub_ops()
        half = world // 2
        cfgp = CfgParallel(rank, world)
        assert (cfgp.stream, cfgp.sp_rank, cfgp.half) == (rank // half, rank % half, half)
        # the half's sequence-parallel subgroup: rank-ordered gathers stay inside the half
        if half > 1:
            got = cfgp.sp.all_gather(torch.tensor([[float(rank)]]))
            assert got.flatten().tolist() == [float(r) for r in range(cfgp.stream * half, (cfgp.stream + 1) * half)]
            assert (cfgp.sp.rank, cfgp.sp.world) == (rank % half, half)
        else:
            assert cfgp.sp is None
        # the swap: (cond, uncond) in stream order on both ranks of the pair {i, i + half}
        c, u = cfgp.exchange(torch.full((2, 3), float(rank)))
        assert c.unique().tolist() == [float(rank % half)] and u.unique().tolist() == [float(rank % half + half)]
        # generate(): the single-process joint pass ...
        ref_model = StreamDiT()
        ref = _generate(WanAny2VHIP(ref_model, device="cpu"))
        assert all(n == 2 for n, _, _ in ref_model.calls)
        # ... and the same generation with the streams on the two halves
        m = StreamDiT()
        pipe = WanAny2VHIP(m, device="cpu")
        pipe.cfg_parallel = cfgp.attach(m)
        out = _generate(pipe)
        assert m.sp is cfgp.sp
        assert len(m.calls) == 5 and all(n == 1 and x_id == cfgp.stream for n, x_id, _ in m.calls)
        assert all(abs(cm - (0.5 if cfgp.stream == 0 else 0.0)) < 1e-6 for _, _, cm in m.calls)   # its own stream's prompt
        assert torch.equal(out, ref), f"max diff {(out - ref).abs().max().item()}"
        # without guidance every rank runs the one forward there is (no swap)
        m1 = StreamDiT()
        p1 = WanAny2VHIP(m1, device="cpu")
        p1.cfg_parallel = cfgp
        a = p1.generate(context=torch.zeros(1, 512, 4096, dtype=torch.bfloat16), width=64, height=64, frame_num=9, sampling_steps=3,
                        guide_scale=1.0, seed=3, return_latents=True)["latents"]
        assert torch.isfinite(a).all() and all(n == 1 and x_id == 0 for n, x_id, _ in m1.calls)
        # a step-skipping cache needs both streams in one process: refused
        import types
        m2 = StreamDiT()
        m2.cache = types.SimpleNamespace(cache_type="mag")
        p2 = WanAny2VHIP(m2, device="cpu")
        p2.cfg_parallel = cfgp
        try:
            _generate(p2)
            raise AssertionError("a cache under CFG parallelism must be refused")
        except NotImplementedError:
            pass
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_cfg_parallel_groups_swap_and_generate(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_cfg_parallel_needs_an_even_world():
    from wan2gp_amd.sp import CfgParallel
    with pytest.raises(ValueError):
        CfgParallel(0, 3)
    with pytest.raises(ValueError):
        CfgParallel(0, 1)
