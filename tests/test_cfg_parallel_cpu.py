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
        # i2v: the pinned start frame is re-noised before every step from the PROCESS-GLOBAL generator (any2video.py:1517-1523), which
        # `seed` does not seed and which differs between ranks (here: seeded with the rank).  Rank 0's draw must reach every rank --
        # otherwise the two halves denoise different latents and the combine mixes predictions of two inputs (round-3 advisor finding)
        class StubVAE:
            def encode(self, videos, tile_size=0, any_end_frame=False):
                out = []
                for v in videos:
                    T, H, W = v.shape[1:]
                    base = torch.nn.functional.adaptive_avg_pool3d(v[None].float(), ((T - 1) // 4 + 1, H // 8, W // 8))[0].mean(0, keepdim=True)
                    out.append(base.repeat(16, 1, 1, 1) + torch.arange(16).view(16, 1, 1, 1) * 0.01)
                return out
        img = torch.rand(3, 64, 64, generator=torch.Generator().manual_seed(9)) * 2 - 1
        mi_ref = StreamDiT(); mi_ref.model_type = "i2v2_2"
        torch.manual_seed(1000)                                  # the state rank 0 starts the parallel run from
        ref_i = _generate(WanAny2VHIP(mi_ref, vae=StubVAE(), device="cpu"), image_start=img)
        mi = StreamDiT(); mi.model_type = "i2v2_2"
        pi = WanAny2VHIP(mi, vae=StubVAE(), device="cpu")
        pi.cfg_parallel = cfgp.attach(mi)
        torch.manual_seed(1000 + rank)
        out_i = _generate(pi, image_start=img)
        assert torch.equal(out_i, ref_i), f"i2v under CFG parallelism: max diff {(out_i - ref_i).abs().max().item()}"
        assert not torch.equal(ref_i, ref)
        # a negative seed means "draw one": rank 0's draw, on every rank -- and the draw leaves this rank's global generator alone
        gstate = torch.get_rng_state()
        drawn = [pi._replicated_seed(-1)]
        assert torch.equal(torch.get_rng_state(), gstate)
        allr = [None] * world
        dist.all_gather_object(allr, drawn[0])
        assert len(set(allr)) == 1
        # two sequence-parallel worlds inside one default group (ranks [0, half) and [half, world)): the latent-sharing group is the
        # SUB-group, its source a global rank -- each group agrees on its own first member's draw and noise (round-4 advisor: a
        # broadcast on WORLD from rank 0 hangs or mixes the groups)
        if half > 1:
            import types
            from wan2gp_amd.sp import SequenceParallel
            groups = [dist.new_group(list(range(s_ * half, (s_ + 1) * half))) for s_ in (0, 1)]
            mine = rank // half
            ms = StreamDiT()
            ms.sp = SequenceParallel(rank % half, half, group=groups[mine])
            ps = WanAny2VHIP(ms, device="cpu")
            grp, src = ps._latent_group()
            assert grp is groups[mine] and src == mine * half
            sd = ps._replicated_seed(-1)
            torch.manual_seed(77 + rank)
            nz = ps._replicated_randn_like(torch.zeros(3, 5))
            got = [None] * world
            dist.all_gather_object(got, (sd, nz.tolist()))
            for g0 in (0, half):
                assert all(got[r_] == got[g0] for r_ in range(g0, g0 + half)), "a sequence-parallel sub-group disagrees on its draw"
            assert got[0][1] != got[half][1]                      # different groups, different noise: nothing crossed the group boundary
        # outside an initialised world nothing is broadcast (cfg_parallel set on a single process: the draw is local)
        # the per-block exchange inside a half can be the Ulysses all-to-alls instead of the all-gathers (bench --parallelism cfg-ulysses)
        cfgu = CfgParallel(rank, world, mode="ulysses")
        assert (cfgu.sp is None) if half == 1 else (cfgu.sp.mode == "ulysses" and cfgu.sp.world == half and cfgu.sp.make_info(64 * half).mode == 1)
        # without guidance every rank runs the one forward there is (no swap)
        m1 = StreamDiT()
        p1 = WanAny2VHIP(m1, device="cpu")
        p1.cfg_parallel = cfgp
        a = p1.generate(context=torch.zeros(1, 512, 4096, dtype=torch.bfloat16), width=64, height=64, frame_num=9, sampling_steps=3,
                        guide_scale=1.0, seed=3, return_latents=True)["latents"]
        assert torch.isfinite(a).all() and all(n == 1 and x_id == 0 for n, x_id, _ in m1.calls)
        # (a step-skipping cache under CFG parallelism -- refused until round 6 -- is served by the unconditional rank making the conditional
        # stream's decision itself: test_step_skipping_decisions_of_the_unconditional_rank_equal_the_single_process_sequence below,
        # tests/test_gpu_skipcache.py::test_cache_on_the_unconditional_rank_of_cfg_parallelism)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
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


def _layout_worker(rank, world, port, q):
    """bench.setup_parallel on a gloo world: the healthy cfg-sp layout, then the agreed fall-back when ONE rank's setup fails."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        import types
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        from wan2gp_amd import sp as SP
        half = world // 2
        m, m2 = types.SimpleNamespace(sp="unset"), types.SimpleNamespace(sp="unset")
        cfgp, cfg_sp, degree, note = bench.setup_parallel(rank, world, True, 75600, (m, m2, None), device="cpu")
        assert cfg_sp and degree == half and note is None and cfgp.stream == rank // half
        assert m.sp is cfgp.sp and m2.sp is cfgp.sp and (cfgp.sp is None) == (half == 1)
        # the swap raises (the same code on every rank, so on every rank -- a failure of ONE rank inside a collective leaves its
        # partner waiting for the backend's timeout, which no in-process fall-back can help): every rank leaves the cfg-sp layout, and
        # a rank that did NOT fail itself follows the others (the self-test of ranks below the last one succeeds in the second pass)
        real = SP.CfgParallel.exchange

        def broken(self, mine):
            raise RuntimeError("injected failure of the swap")
        SP.CfgParallel.exchange = broken
        m, m2 = types.SimpleNamespace(sp="unset"), types.SimpleNamespace(sp="unset")
        cfgp, cfg_sp, degree, note = bench.setup_parallel(rank, world, True, 75600, (m, m2, None), device="cpu")
        assert cfgp is None and not cfg_sp and degree == world
        assert isinstance(m.sp, SP.SequenceParallel) and m.sp is m2.sp and (m.sp.rank, m.sp.world, m.sp.group) == (rank, world, None)
        assert "fell back" in note and "injected failure" in note
        got = m.sp.all_gather(torch.tensor([[float(rank)]]))           # the fall-back layout's group works: the whole world
        assert got.flatten().tolist() == [float(r) for r in range(world)]
        # a rank whose own self-test passed still follows a rank that reports a failure (here: a wrong result seen by the last rank only,
        # after the swap completed everywhere)
        def wrong_on_last(self, mine):
            a, b = real(self, mine)
            return (a + 1, b) if self.rank == world - 1 else (a, b)
        SP.CfgParallel.exchange = wrong_on_last
        m = types.SimpleNamespace(sp="unset")
        cfgp, cfg_sp, degree, note = bench.setup_parallel(rank, world, True, 75600, (m,), device="cpu")
        assert cfgp is None and not cfg_sp and degree == world and m.sp.world == world
        assert ("another rank" in note) == (rank != world - 1)
        SP.CfgParallel.exchange = broken
        # asked for explicitly, or a token count the whole world does not shard: fatal on every rank, not a silent change of layout
        for demanded, L in ((True, 75600), (False, 75600 + half)):
            try:
                bench.setup_parallel(rank, world, True, L, (m,), demanded, device="cpu")
                raise AssertionError("must exit")
            except SystemExit as ex:
                assert "cfg-sp" in str(ex.code)
        # the Ulysses exchange (the bench's default where the heads divide by the degree) is self-tested the same way: healthy -> kept;
        # a failing all-to-all -> EVERY rank keeps the K / V^T all-gathers and says so; asked for explicitly -> fatal
        SP.CfgParallel.exchange = real
        m = types.SimpleNamespace(sp="unset")
        cfgp, cfg_sp, degree, note = bench.setup_parallel(rank, world, True, 75600, (m,), device="cpu", sp_mode="ulysses")
        assert cfg_sp and note is None and ((cfgp.sp is None) if half == 1 else cfgp.sp.mode == "ulysses")
        cfgp, cfg_sp, degree, note = bench.setup_parallel(rank, world, False, 75600, (m,), device="cpu", sp_mode="ulysses")
        assert cfgp is None and note is None and m.sp.mode == "ulysses" and m.sp.world == world
        real_a2a = dist.all_to_all_single

        def broken_a2a(*a, **k):
            raise RuntimeError("injected failure of the all-to-all")
        dist.all_to_all_single = broken_a2a
        cfgp, cfg_sp, degree, note = bench.setup_parallel(rank, world, False, 75600, (m,), device="cpu", sp_mode="ulysses")
        assert m.sp.mode == "allgather" and "self-test failed" in note and "injected failure" in note
        try:
            bench.setup_parallel(rank, world, False, 75600, (m,), device="cpu", sp_mode="ulysses", sp_mode_demanded=True)
            raise AssertionError("must exit")
        except SystemExit as ex:
            assert "asked for" in str(ex.code)
        dist.all_to_all_single = real_a2a
        # plain sequence parallelism when cfg-sp was not selected (odd worlds, --parallelism sp)
        SP.CfgParallel.exchange = real
        m = types.SimpleNamespace(sp=None)
        assert bench.setup_parallel(rank, world, False, 75600, (m,), device="cpu")[:3] == (None, False, world) and m.sp.world == world
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_layout_falls_back_to_sequence_parallelism_on_every_rank_together(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_layout_worker, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=240) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _hang_worker(rank, world, port, q, never_wakes):
    """bench.setup_parallel with ONE rank that never enters the Ulysses all-to-all (round 6, first-run hardening): the others must not
    sit in the collective for the backend's timeout -- every rank gives the exchange up after the guard, all agree on the all-gather
    fall-back over the side channel, the fall-back runs on a FRESH group (the old one holds the unmatched all-to-all) and works."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.update(WAN_BENCH_GUARD_S="2", WAN_BENCH_AGREE_S="6" if never_wakes else "30", WAN_BENCH_INJECT_A2A_HANG_RANK=str(world - 1),
                      WAN_BENCH_INJECT_A2A_HANG_S="1000" if never_wakes else "4")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        import time
        import types
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        assert bench.GUARD_S == 2.0
        bench.make_agreement_group()
        m = types.SimpleNamespace(sp="unset")
        t0 = time.perf_counter()
        if never_wakes:
            # the hung rank never reaches the agreement either: the others cannot decide together -> WorldLost on each of them, inside
            # guard + agreement time (main() then leaves the world: rank 0 alone on one GPU)
            if rank == world - 1:
                q.put((rank, "ok"))
                time.sleep(20)                      # (stays alive: its sockets stay open, as a hung rank's would)
                return
            try:
                bench.setup_parallel(rank, world, False, 75600, (m,), device="cpu", sp_mode="ulysses")
                raise AssertionError("must raise WorldLost")
            except bench.WorldLost as ex:
                assert "could not agree" in str(ex)
            assert time.perf_counter() - t0 < 15
            q.put((rank, "ok"))
            return
        cfgp, cfg_sp, degree, note = bench.setup_parallel(rank, world, False, 75600, (m,), device="cpu", sp_mode="ulysses")
        assert time.perf_counter() - t0 < 25
        assert cfgp is None and degree == world and m.sp.mode == "allgather" and m.sp.group is not None      # a fresh group
        assert "hung" in note and ("this rank" in note) == True or "another rank" in note
        got = m.sp.all_gather(torch.tensor([[float(rank)]]))
        assert got.flatten().tolist() == [float(r) for r in range(world)]
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if not never_wakes:
            dist.destroy_process_group()


@pytest.mark.parametrize("world,never_wakes", [(2, False), (4, False), (4, True)])
def test_a_rank_hanging_in_the_ulysses_self_test_demotes_the_layout_on_every_rank(world, never_wakes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hang_worker, args=(r, world, port, q, never_wakes), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=120) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30 if not never_wakes else 1)
            if p.is_alive():
                p.kill()
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.parametrize("kind,x_count", [("tea", 2), ("mag", 2), ("mag", 3)])
def test_step_skipping_decisions_of_the_unconditional_rank_equal_the_single_process_sequence(kind, x_count):
    """A step-skipping cache under CFG parallelism (refused until round 6).  In one process the reference decides for the conditional pass
    (x_id 0) and the unconditional pass (x_id 1) of a step in turn; TeaCache's x_id-1 pass and MagCache's one_for_all form (x_count > 2) read
    the x_id-0 verdict (model.py:1921-1923, :1945-1946).  That verdict depends on the timestep's embedding and the cache's counters only:
    the rank that runs stream 1 makes it itself first (WanModelHIP.forward, cfg_parallel_stream) -- same flags for both streams as the
    single-process order, for every step, with nothing exchanged."""
    import numpy as np
    import torch
    from wan2gp_amd import skipcache as SC
    steps = 12
    g = torch.Generator().manual_seed(5)
    base = torch.randn(1, 64, generator=g)
    es = [base * (1 + 0.05 * i) for i in range(steps)]            # relative L1 between neighbours ~ 0.04: three skips, then a computed step

    def mk():
        c = SC.SkipStepsCache(cache_type=kind, multiplier=2.0, start_step=1, num_steps=steps, skipped_steps=0, previous_residual=None,
                              previous_modulated_input=None)
        if kind == "mag":
            c.update({"magcache_thresh": 0.08, "magcache_K": 3, "mag_ratios": np.concatenate([[1.0, 1.0], 1.0 - 0.02 * np.arange(1, 2 * steps - 1) / steps])})
        else:
            c.update({"coefficients": [1.0, 0.0], "rel_l1_thresh": 0.15, "accumulated_rel_l1_distance": 0})
        SC.reset_for_generation(c, x_count)
        return c
    one, r0, r1 = mk(), mk(), mk()
    want, got = [], []
    for i in range(steps):
        e = es[i] if kind == "tea" else None
        want.append((SC.decide(one, 1, 0, i, e)[0], SC.decide(one, 1, 1, i, None)[0]))
        f0 = SC.decide(r0, 1, 0, i, e)[0]                       # the conditional rank: its own pass
        SC.decide(r1, 1, 0, i, e)                               # the unconditional rank: stream 0's decision first ...
        got.append((f0, SC.decide(r1, 1, 1, i, None)[0]))       # ... then its own pass
    assert got == want
    assert any(not a for a, _ in want) and any(a for a, _ in want[2:])      # the scenario skips some steps and computes others
    assert r1.skipped_steps == one.skipped_steps == r0.skipped_steps
