"""CPU: the host half of `wan_sched_*` (no device work): `wan_sched_set_timesteps` must produce exactly the timesteps and sigmas
of the reference's schedulers (tests/golden/sched.npz: fm_solvers_unipc.py set_timesteps, euler_scheduler.py set_timesteps).
The stepping itself launches kernels and is covered by tests/test_gpu_model.py::test_native_*."""
import os
from ctypes import byref, c_double, c_float, c_void_p

import numpy as np
import pytest

from wan2gp_amd import lib as L

G = os.path.join(os.path.dirname(__file__), "golden", "sched.npz")


def make(kind):
    h = c_void_p()
    L.check(L.load().wan_sched_create(byref(h), kind, 1000), "create")
    return h


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (30, 12.0), (4, 3.0)])
def test_unipc_timesteps_and_sigmas_equal_the_reference(steps, shift):
    g = np.load(G)
    h = make(0)
    ts, sg = (c_double * steps)(), (c_float * (steps + 1))()
    L.check(L.load().wan_sched_set_timesteps(h, steps, shift, ts, sg), "set_timesteps")
    assert np.array_equal(np.array(list(ts), dtype=np.int64), g[f"unipc_ts_{steps}_{shift}"])
    assert np.array_equal(np.array(list(sg), dtype=np.float32), g[f"unipc_sig_{steps}_{shift}"])
    L.load().wan_sched_destroy(h)


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (4, 3.0)])
def test_euler_timesteps_equal_the_reference(steps, shift):
    g = np.load(G)
    h = make(1)
    ts = (c_double * steps)()
    L.check(L.load().wan_sched_set_timesteps(h, steps, shift, ts, None), "set_timesteps")
    assert np.array_equal(np.array(list(ts), dtype=np.float32), g[f"euler_ts_{steps}_{shift}"])
    L.load().wan_sched_destroy(h)


def test_bad_arguments_are_errors_not_crashes():
    h = c_void_p()
    assert L.load().wan_sched_create(byref(h), 7, 1000) != 0 and b"kind" in L.load().wan_last_error()
    h = make(0)
    assert L.load().wan_sched_set_timesteps(h, 0, 5.0, None, None) != 0
    assert L.load().wan_sched_step(h, None, 0.0, None, None, 0, None) != 0
    L.load().wan_sched_destroy(h)
    L.load().wan_sched_destroy(None)
