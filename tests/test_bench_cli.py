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
