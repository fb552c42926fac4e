"""CPU: bench.py's launcher contract -- `python bench.py --gpus N` (N > 1) outside torch.distributed.run re-executes itself
as N ranks on 127.0.0.1 (the way the driver launches multi-GPU runs); under torch.distributed.run it must not spawn again."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import importlib
    import bench
    return importlib.reload(bench)


def test_gpus_n_self_spawns_one_rank_per_gpu(monkeypatch):
    bench = _bench()
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: (seen.update(cmd=cmd, env=env), 0)[1])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_is_an_error(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: pytest.fail("must not spawn under torch.distributed.run"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_workload_flops_match_the_survey_table():
    bench = _bench()
    cfg, (f, h, w), _ = bench.WORKLOADS["14B-720p"]
    L = f * (h // 2) * (w // 2)
    assert L == 75600 and abs(bench.forward_flops(cfg, L) / 6.52e15 - 1) < 5e-3          # SURVEY.md section 8 shape table
    cfg, (f, h, w), _ = bench.WORKLOADS["1.3B-480p"]
    assert abs(bench.forward_flops(cfg, f * (h // 2) * (w // 2)) / 2.83e14 - 1) < 5e-3


def test_more_ranks_than_gpus_is_an_error_before_any_collective(monkeypatch):
    """torch.distributed.run started N ranks on a node with fewer GPUs: every rank exits with a message instead of hanging in
    init_process_group / sharing a device."""
    bench = _bench()
    import torch
    monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: pytest.fail("must stop before touching a device"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "4 ranks need 4 GPUs" in str(e.value.code)


@pytest.mark.parametrize("world,parallelism,degree", [(32, "sp", 32), (64, "auto", 32), (64, "cfg-sp", 32)])
def test_token_count_that_does_not_shard_is_an_error_before_the_weights_are_built(monkeypatch, world, parallelism, degree):
    """L = 75,600 tokens shard over 2 / 4 / 8 / 16 sequence-parallel ranks; a degree that does not divide them (32) stops with a
    message that names the numbers, not with a ValueError inside the first forward of one rank while the others wait.  Under
    cfg-sp (the default for an even world) the degree is HALF the world: the two halves run the two CFG streams."""
    bench = _bench()
    import torch
    monkeypatch.setenv("WORLD_SIZE", str(world)); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(world), "--parallelism", parallelism])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: world)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: pytest.fail("must stop before touching a device"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "75600" in str(e.value.code) and f"over {degree} sequence-parallel" in str(e.value.code)


def test_auto_layout_follows_the_link_modelled_tables():
    """choose_layout (round 5): N = 2 -> cfg2 x sp1, N = 4 -> cfg2 x sp2 over the all-gathers, N = 8 with 40 heads -> the token axis over all 8
    ranks with the Ulysses exchange (every all-to-all over 7 links), 12 heads (1.3B) -> cfg2 x sp4 (ulysses), N = 16 -> cfg2 x sp8 (ulysses);
    an odd world is plain sequence parallelism; explicit modes are taken literally."""
    b = _bench()
    assert b.choose_layout(1, "auto", 40) == (False, 1, "allgather")
    assert b.choose_layout(2, "auto", 40) == (True, 1, "allgather")
    assert b.choose_layout(4, "auto", 40) == (True, 2, "allgather")
    assert b.choose_layout(8, "auto", 40) == (False, 8, "ulysses")
    assert b.choose_layout(8, "auto", 12) == (True, 4, "ulysses")
    assert b.choose_layout(16, "auto", 40) == (True, 8, "ulysses")
    assert b.choose_layout(3, "auto", 40) == (False, 3, "allgather")
    assert b.choose_layout(64, "auto", 40) == (True, 32, "allgather")
    assert b.choose_layout(8, "cfg-ulysses", 40) == (True, 4, "ulysses") and b.choose_layout(8, "cfg-sp", 40) == (True, 4, "allgather")
    assert b.choose_layout(8, "sp", 40) == (False, 8, "allgather") and b.choose_layout(8, "ulysses", 40) == (False, 8, "ulysses")


def test_cfg_sp_needs_an_even_world(monkeypatch):
    bench = _bench()
    import torch
    monkeypatch.setenv("WORLD_SIZE", "3"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--parallelism", "cfg-sp"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 3)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: pytest.fail("must stop before touching a device"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "odd" in str(e.value.code)


def test_committed_bench_line_carries_the_contract():
    """The bench line measured on the committed code (profiles/r03_bench_14B-720p_run87.json, written by `python bench.py` on an MI355X):
    every field of the driver's contract, the roofline and cpu_baseline objects of the tier framing, self-consistent numbers."""
    import json
    p = os.path.join(ROOT, "profiles", "r03_bench_14B-720p_run87.json")
    j = json.load(open(p))
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                   ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(j[k], typ), k
    assert "vs_baseline" in j and j["vs_baseline"] is None           # BASELINE.md holds no published number for this metric
    assert j["metric"] == "denoise-steps/s" and j["higher_is_better"] is True and j["n_gpus"] == 1 and "workload" in j["config"]
    assert abs(j["value"] * j["ms_per_step"] / 1000.0 - 1.0) < 1e-6  # steps/s x s/step
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] > 6.19e9 and "pmc" in r["traffic_source"].lower()      # PMC bytes per launch >= the algorithmic bytes
    assert r["declined_workgroups"] == 0 and r["total_workgroups"] == 80 * 23680
    assert 0.5 < r["frac_of_sustained_mfma"] < 1.0 and r["sustained_mfma"]["TFLOPs"] > 1500
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "denoise-steps/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["cpu_model"]
    assert [x["world"] for x in j["simulated_scaling"]["ranks"]] == [2, 4, 8]
    assert all(0.5 < x["compute_side_efficiency"] <= 1.0 for x in j["simulated_scaling"]["ranks"])
    assert j["config5"]["dtype"].startswith("fp8") and j["secondary"]["ms_per_step"] > 0 and j["e2e"]["composed_s_at_30_steps"] > 0


def test_optional_blocks_are_skipped_past_the_extras_budget_and_failures_are_recorded(monkeypatch):
    """_extra_block: an optional block never costs the JSON line -- it is skipped (with a note) once the process is older than the
    budget, a failing one leaves its error in its place, one inside the budget runs."""
    import bench
    calls = []
    monkeypatch.setattr(bench, "T_PROCESS0", bench.time.perf_counter() - 1000.0)
    r = bench._extra_block(lambda: calls.append(1) or {"ok": 1}, budget_s=900.0)
    assert "skipped" in r and not calls
    assert bench._extra_block(lambda: {"ok": 1}, budget_s=1100.0) == {"ok": 1}
    assert bench._extra_block(lambda: {"ok": 1}) == {"ok": 1}                      # no budget: always runs

    def boom():
        raise RuntimeError("no")
    r = bench._extra_block(boom, budget_s=1100.0)
    assert "RuntimeError" in r["error"] and "traceback" in r


def test_simulated_cfg_sp_rank_runs_one_stream_and_leaves_the_models_as_they_came(monkeypatch):
    """simulate_world(layout='cfg-sp'), world 2 (no sequence parallelism inside a half, so no device stream is needed): the simulated
    rank sends ONE stream through the model, the swap is a copy, the row says so; an odd world is skipped; the stand-in and the
    models' `sp` are gone afterwards."""
    bench = _bench()
    import torch
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    calls = []

    class M:
        sp = None

        def __call__(self, x, context, **kw):
            calls.append((len(x), kw.get("x_id")))
            return [torch.zeros(2) for _ in x]
    m = M()
    par = {"cfgp": None}

    def one_step(i, lat, sc=None, fr=None):
        c, u = par["cfgp"].guided_pair(m, lat, "ctx", "null")
        assert c is not u and torch.equal(c, u)
        return lat
    r = bench.simulate_world([2, 3], m, None, one_step, torch.zeros(2), lambda: None, 8.0, {"num_layers": 40}, 75600, par, "cfg-sp", link_GBs=0.0)
    rows = r["ranks"]
    assert rows[0]["layout"] == "cfg2 x sp1" and rows[0]["streams_per_rank"] == 1 and rows[0]["tokens_per_rank"] == 75600
    assert rows[0]["gathered_bytes_per_block_and_rank"] == 0.0 and "skipped" in rows[1] and "rank_step_ms_link" not in rows[0]
    assert calls == [(1, 0)] * 3 and par["cfgp"] is None and m.sp is None
    # the link model (round 5): the same steps once more with the swap's transfer time behind the copy (19 MB-sized in the bench; here 8 bytes)
    delays = []
    monkeypatch.setattr(bench, "_link_delay", lambda nbytes, rate: delays.append((nbytes, rate)))
    r = bench.simulate_world([2], m, None, one_step, torch.zeros(2), lambda: None, 8.0, {"num_layers": 40}, 75600, par, "cfg-sp", link_GBs=50.0)
    row = r["ranks"][0]
    assert r["link_model_GBs_per_peer"] == 50.0 and {"rank_step_ms_link", "link_modelled_efficiency", "exposed_ms_per_block"} <= set(row)
    assert delays == [(8, 50.0)] * 2 and len(calls) == 3 + 5                 # warm-up + 2 compute-only steps + 2 link-modelled steps


class _StrictScheduler:
    """The native scheduler's contract that cost round 3's driver run its scaling table: `n` timesteps, and stepping past the last
    one raises ("wan_sched_step: stepped past the last timestep")."""
    made = 0

    def __init__(self, n=30):
        import torch
        type(self).made += 1
        self.timesteps = torch.arange(999, 999 - n, -1)
        self.n, self.used = n, 0

    def step(self, noise, t, lat):
        if self.used >= self.n:
            raise RuntimeError("wan_sched_step: stepped past the last timestep")
        self.used += 1
        return (lat,)


def test_blocks_behind_the_timed_region_do_not_depend_on_the_timesteps_it_consumed(monkeypatch):
    """The driver's own command line, `--gpus 1 --steps 20 --warmup 5`, leaves 5 of the main scheduler's 30 timesteps; round 3's
    simulated-ranks block kept stepping THAT scheduler (6 plans x 3 steps) and died.  Every block now takes a scheduler of its own
    from `new_sched`: the same sequence on a scheduler with the native one's refusal -- 25 steps of the timed region, then worlds
    2 / 4 / 8 in both layouts, then the 161-frame configuration's world of 8 -- returns data in every row."""
    bench = _bench()
    import torch
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: object())
    _StrictScheduler.made = 0
    main_sched = _StrictScheduler(30)
    par = {"cfgp": None}

    class M:
        sp = None
    m, m2 = M(), M()
    seen = []

    def one_step(i, lat, sc=None, fr=None):
        sc = main_sched if sc is None else sc
        t = sc.timesteps[i]
        seen.append((sc is main_sched, m.sp.world if m.sp is not None else 1, par["cfgp"] is not None, fr))
        return sc.step(None, t, lat)[0]
    lat = torch.zeros(2)
    for i in range(25):                                   # --warmup 5 --steps 20
        lat = one_step(i, lat)
    r = bench.simulate_world([2, 4, 8], m, m2, one_step, lat, _StrictScheduler, 8.5, {"num_layers": 40}, 75600, par, "both", link_GBs=0.0)
    rows = r["ranks"]
    assert [(x["world"], x["layout"]) for x in rows] == [(2, "sp2"), (2, "cfg2 x sp1"), (4, "sp4"), (4, "cfg2 x sp2"), (8, "sp8"), (8, "cfg2 x sp4")]
    assert all("rank_step_ms" in x and "error" not in x and "skipped" not in x for x in rows)
    assert [x["tokens_per_rank"] for x in rows] == [37800, 75600, 18900, 37800, 9450, 18900]
    assert _StrictScheduler.made == 1 + 2 * 6 and main_sched.used == 25          # per plan: the warm-up's scheduler and the timed run's
    assert m.sp is None and m2.sp is None and par["cfgp"] is None
    # the simulated steps saw their own scheduler, the shard's world and (cfg-sp) the stand-in for the 2-rank swap
    sim = seen[25:]
    assert len(sim) == 18 and not any(s[0] for s in sim)
    assert [s[1] for s in sim[::3]] == [2, 1, 4, 2, 8, 4] and [s[2] for s in sim[::3]] == [False, True] * 3
    # BASELINE configs[3] (161 frames, L = 147,600): the same block at that L with its own rope tables handed through
    r3 = bench.simulate_world([8], m, m2, one_step, lat, _StrictScheduler, 28.0, {"num_layers": 40}, 147600, par, "both", fr="freqs161", link_GBs=0.0)
    assert [x["tokens_per_rank"] for x in r3["ranks"]] == [18450, 36900] and all(s[3] == "freqs161" for s in seen[43:])
    # all four layouts (the default of the bench line): the Ulysses rows carry the all-to-all exchange, heads must divide by the degree,
    # cfg2 x sp1 has no exchange and therefore no Ulysses twin
    modes = []

    def one_step2(i, lat, sc=None, fr=None):
        modes.append((m.sp.mode, m.sp.world) if m.sp is not None else None)
        return sc.step(None, sc.timesteps[i], lat)[0]
    r4 = bench.simulate_world([2, 8], m, m2, one_step2, lat, _StrictScheduler, 8.5, {"num_layers": 40, "num_heads": 40}, 75600, par, "all", None, 1, 0.0)
    assert [x["layout"] for x in r4["ranks"]] == ["sp2", "cfg2 x sp1", "sp2 (ulysses)", "sp8", "cfg2 x sp4", "sp8 (ulysses)", "cfg2 x sp4 (ulysses)"]
    assert [x["exchange"].split(" ")[0].rstrip(":") for x in r4["ranks"]] == ["all-gather", "none", "all-to-all", "all-gather", "all-gather", "all-to-all", "all-to-all"]
    assert all("2 head chunks" in x["exchange"] for x in r4["ranks"] if "ulysses" in x["layout"])     # 20 / 5 / 10 heads per rank: the library default
    assert modes[::2] == [("allgather", 2), None, ("ulysses", 2), ("allgather", 8), ("allgather", 4), ("ulysses", 8), ("ulysses", 4)]
    # the link model on: a Ulysses row runs chunked AND as one exchange per tensor, both with and without the model, and the two must agree bit for bit
    modes.clear()
    chunk_seen = []

    def one_step3(i, lat, sc=None, fr=None):
        chunk_seen.append((m.sp.link_GBs, m.sp.resolved_chunks(40)))
        return sc.step(None, sc.timesteps[i], lat)[0]
    r6 = bench.simulate_world([8], m, m2, one_step3, lat, _StrictScheduler, 8.5, {"num_layers": 40, "num_heads": 40}, 75600, par, "cfg-ulysses", None, 1, 50.0)
    row = r6["ranks"][0]
    assert chunk_seen == [(0.0, 2), (0.0, 2), (50.0, 2), (0.0, 1), (50.0, 1)]                            # warm-up, then the four figures
    assert row["one_exchange"]["latents_bit_identical_to_chunked"] is True and row["one_exchange"]["latents_max_abs_diff_to_chunked"] == 0.0
    assert "chunking_gain_points" in row and "exposed_ms_per_block" in row
    r5 = bench.simulate_world([8], m, m2, one_step2, lat, _StrictScheduler, 0.4, {"num_layers": 30, "num_heads": 12}, 32760, par, "ulysses")
    assert "12 heads do not divide by 8" in r5["ranks"][0]["skipped"]


def test_the_161_frame_workload_is_baseline_configs_3():
    """BASELINE.json configs[3]: 720p x 161 frames -> (161 - 1) // 4 + 1 = 41 latent frames (any2video.py:647), 41 x 45 x 80 = 147,600
    tokens, which shard over 2 / 4 / 8 sequence-parallel ranks and over the 4-rank halves of cfg2 x sp4."""
    bench = _bench()
    cfg, (f, h, w), desc = bench.WORKLOADS["14B-720p-161f"]
    assert f == (161 - 1) // 4 + 1 and (h * 8, w * 8) == (720, 1280)
    L = f * (h // 2) * (w // 2)
    assert L == 147600 and all(L % n == 0 for n in (2, 4, 8)) and "161f" in desc
    assert cfg == bench.WORKLOADS["14B-720p"][0] and "14B-720p-161f" in bench.TWO_EXPERT_WORKLOADS


def test_cpu_baseline_reports_its_thread_sweep():
    """cpu_baseline(): the thread sweep is in the record, the full sample ran at the best count of each leg (a tiny stand-in for the
    1.3B model so that the test takes seconds).  In a process of its own: torch.set_num_threads changes which reduction order later
    CPU kernels pick, and the oracle-vs-golden tests of this suite compare bit patterns."""
    import json
    import subprocess
    code = ("import json, sys; sys.path.insert(0, %r); import bench, torch; "
            "r = bench.cpu_baseline(1.0e16, sweep=(1, 2), _cfg_name='tiny', _fhw=(2, 8, 8)); "
            "r['threads_after'] = torch.get_num_threads(); print('RESULT ' + json.dumps(r))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert r["kind"] == "port" and r["cores"] in (1, 2, r["torch_default_threads"]) and r["value"] > 0
    assert set(r["thread_sweep"]["dit_3_layers_s"]) >= {"1", "2"} and set(r["thread_sweep"]["vae_first_frame_s"]) >= {"1", "2"}
    assert str(r["cores"]) in r["thread_sweep"]["dit_3_layers_s"] and "threads" in r["sample"]
    assert r["threads_after"] == r["torch_default_threads"]                       # restored


def test_the_drivers_own_command_line_produced_the_whole_record():
    """profiles/r04_bench_14B-720p_run11.json = `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command, which cost round 3
    its scaling table) on an MI355X: every block behind the timed region returned data -- the simulated layouts in all four forms at
    worlds 2 / 4 / 8, BASELINE configs[3] with its own world of 8, the attention robustness probe, config 5, the CPU baseline with its
    thread sweep -- and the numbers are self-consistent."""
    import json
    for name in ("r04_bench_14B-720p_run11.json", "r04_bench_14B-720p_run30.json"):     # mid-round (slowest box met) and the round's last tree (a fast one)
        _check_driver_line(json.load(open(os.path.join(ROOT, "profiles", name))))


def _check_driver_line(j):
    import json
    assert (j["steps"], j["warmup"], j["n_gpus"]) == (20, 5, 1) and abs(j["value"] * j["ms_per_step"] / 1000.0 - 1.0) < 1e-6
    assert "error" not in json.dumps({k: j[k] for k in ("secondary", "simulated_scaling", "configs3", "config5", "cpu_baseline")}).lower().replace("max_abs_err", "")
    rows = j["simulated_scaling"]["ranks"]
    assert [(r["world"], r["layout"]) for r in rows] == [(2, "sp2"), (2, "cfg2 x sp1"), (2, "sp2 (ulysses)"), (4, "sp4"), (4, "cfg2 x sp2"), (4, "sp4 (ulysses)"),
                                                         (4, "cfg2 x sp2 (ulysses)"), (8, "sp8"), (8, "cfg2 x sp4"), (8, "sp8 (ulysses)"), (8, "cfg2 x sp4 (ulysses)")]
    assert all(0.8 < r["compute_side_efficiency"] < 1.05 for r in rows)
    by = {r["layout"]: r for r in rows}
    assert by["cfg2 x sp4 (ulysses)"]["gathered_bytes_per_block_and_rank"] < 0.55 * by["cfg2 x sp4"]["gathered_bytes_per_block_and_rank"]
    assert by["cfg2 x sp4 (ulysses)"]["compute_side_efficiency"] > by["cfg2 x sp4"]["compute_side_efficiency"] > by["sp8"]["compute_side_efficiency"]
    c3 = j["configs3"]
    assert c3["tokens"] == 147600 and c3["steps"] == 2 and 0.5 < c3["roofline"]["frac"] < 0.7 and len(c3["simulated_scaling"]["ranks"]) == 4
    assert abs(c3["roofline"]["achieved"] - c3["roofline"]["flop_per_launch"] / (c3["roofline"]["avg_ms"] * 1e-3) / 1e12) < 1e-6 * c3["roofline"]["achieved"]
    r = j["roofline"]
    rb = r["robustness"]
    assert rb["gain_12_shifted_loop"]["reached_tracking_loop_frac"] == 0.0 and rb["gain_12_shifted_loop"]["TFLOPs"] >= 0.95 * rb["gain_1_plain_loop"]["TFLOPs"]
    assert rb["adversarial_key_all_redone_by_tracking_loop"]["reached_tracking_loop_frac"] == 1.0 and all(v["finite"] for v in rb.values())
    assert r["frac_gain_12"] == rb["gain_12_shifted_loop"]["frac"] and r["frac_all_declined"] == rb["adversarial_key_all_redone_by_tracking_loop"]["frac"]
    assert r["declined_workgroups"] == 0 and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-9
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and str(c["cores"]) in c["thread_sweep"]["dit_3_layers_s"] and len(c["thread_sweep"]["dit_3_layers_s"]) >= 4
    assert min(c["thread_sweep"]["dit_3_layers_s"].values()) == c["thread_sweep"]["dit_3_layers_s"][str(c["cores"])]
    assert j["config5"]["dtype"].startswith("fp8") and j["config5"]["ms_per_step"] < j["ms_per_step"]


def test_giving_the_world_up_leaves_rank_0_as_a_single_gpu_run(monkeypatch):
    """Round 6, first-run hardening: when the ranks cannot run any multi-GPU layout (WorldLost, or the deadline), ranks > 0 exit with
    status 0 -- a non-zero exit would make torch.distributed.run kill rank 0 -- and rank 0 re-executes itself as a short single-GPU
    bench, outside the rendezvous, whose JSON line carries what happened."""
    import json
    bench = _bench()
    seen = {}
    monkeypatch.setattr(bench.os, "_exit", lambda code: (_ for _ in ()).throw(SystemExit(code)))
    monkeypatch.setattr(bench.os, "execve", lambda exe, cmd, env: seen.update(exe=exe, cmd=cmd, env=env))
    with pytest.raises(SystemExit) as e:
        bench._leave_world(3, 8, ["--workload", "14B-720p", "--steps", "5", "--warmup", "1"], "the all-gather self-test hung")
    assert e.value.code == 0 and not seen
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("MASTER_PORT", "1234"); monkeypatch.setenv("LOCAL_RANK", "0")
    bench._leave_world(0, 8, ["--workload", "14B-720p", "--steps", "5", "--warmup", "1"], "the all-gather self-test hung")
    cmd, env = seen["cmd"], seen["env"]
    assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py") and cmd[cmd.index("--gpus") + 1] == "1"
    assert cmd[cmd.index("--steps") + 1] == "5" and "--no-cpu-baseline" in cmd and cmd[cmd.index("--simulate-world") + 1] == ""
    assert not any(k in env for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK"))
    fb = json.loads(env["WAN_BENCH_FELL_BACK"])
    assert fb == {"requested_gpus": 8, "reason": "the all-gather self-test hung"}


def test_the_world_deadline_fires_once_and_can_be_disarmed():
    import time
    bench = _bench()
    fired = []
    d = bench._Deadline(time.perf_counter() - bench.T_PROCESS0 + 1.2, fired.append)
    d.start()
    d.join(5)
    assert len(fired) == 1 and "WAN_BENCH_WORLD_DEADLINE_S" in fired[0]
    fired2 = []
    d = bench._Deadline(time.perf_counter() - bench.T_PROCESS0 + 1.2, fired2.append)
    d.start()
    d.disarm()
    d.join(5)
    assert not fired2


def test_guarded_calls_report_hangs_errors_and_results():
    import time
    bench = _bench()
    assert bench._guarded(lambda: 7, 2) == (True, 7)
    done, r = bench._guarded(lambda: 1 / 0, 2)
    assert done and isinstance(r, ZeroDivisionError) and bench._outcome(done, r)[0] == 1
    done, r = bench._guarded(lambda: time.sleep(30), 0.3)
    assert not done and bench._outcome(done, r)[0] == 2 and "hung" in bench._outcome(done, r)[1]
