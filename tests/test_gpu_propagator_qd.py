"""The covariance an IMU step of the C++ mirror propagates when the discrete process noise is the REFERENCE's -- its numbers
are the fixture tests/golden/propagator_qd.npz (generated from the reference's own scalar statements, propagator.cpp:207-840) --
handed over the way a drop-in does it, Propagator::setProcessNoiseFunction; and the state ring wrapping with the covariance
resident on the device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN_DIR, rel
from oracle import ref_np

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "x_multi_agent_amd")
c_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(c_dp)


def test_covariance_step_with_the_references_qd(xk):
    host = C.CDLL(os.path.join(PKG, "libx_host.so"))
    g = np.load(os.path.join(GOLDEN_DIR, "propagator_qd.npz"))
    N, M = 4, 2
    n = 15 + 6 * N + 3 * M
    rng = np.random.default_rng(11)
    worst = 0.0
    for i in range(len(g["dt"])):
        dt = float(g["dt"][i])
        A = rng.standard_normal((n, n))
        P = np.asfortranarray(A @ A.T * 1e-3 + np.eye(n) * 1e-2)
        s0 = np.zeros(23); s1 = np.zeros(23)
        s0[0], s1[0] = 5.0, 5.0 + dt
        s0[7:11] = s1[7:11] = g["q"][i]
        s1[17:20], s1[20:23] = g["e_w"][i], g["e_a"][i]           # biases zero: the unbiased measurements ARE e_w, e_a
        Qd = np.asfortranarray(g["Q"][i])
        Pout, Fd, calls = np.zeros((n, n), order="F"), np.zeros((15, 15), order="F"), C.c_int(0)
        rc = host.x_host_propagate_covariance_with_qd(_p(s0), _p(s1), _p(Qd), _p(P), C.c_int(N), C.c_int(M), _p(Pout), _p(Fd), C.byref(calls))
        assert rc == 0 and calls.value >= 1
        assert rel(Fd, g["F"][i]) <= 1e-14
        ref = ref_np.propagate_covariance_matrices(P, g["F"][i], g["Q"][i])
        worst = max(worst, rel(Pout, ref))
        assert rel(Pout, ref) <= 1e-13, i
        # and it is NOT what the clean model gives: the fixture matters
        n_w, n_bw, n_a, n_ba = g["noise"][i]
        alt = ref_np.propagate_covariance_matrices(P, g["F"][i], ref_np.process_noise_model(dt, g["e_w"][i], g["e_a"][i], g["q"][i], n_w, n_bw, n_a, n_ba))
        assert np.abs(alt - ref).max() > 0
    print("worst rel dP with the reference's q_d injected:", worst)


@pytest.mark.parametrize("n_steps,bsz", [(3, 6), (5, 6), (23, 6), (40, 9)])
def test_ring_wraps_onto_the_resident_covariance(xk, n_steps, bsz):
    exe = os.path.join(PKG, "xk_ring_wrap_example")
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(n_steps), str(bsz)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("bsz,expect", [(6, "OK discarded"), (9, "OK discarded"), (40, "OK threw")])
def test_ring_wrapping_during_an_update_discards_that_update_like_the_reference(xk, bsz, expect):
    """buffer_sz - 1 IMU samples arrive WHILE an update runs on the resident covariance (the update runs without the
    Ekf's mutex, ekf.cpp:186-205).  The reference overwrites the slot, warns and loses that update (ekf.cpp:229-239); the mirror now
    does the same -- nullopt, the prior restored on the device and carried past the overwritten slots -- and ends with the same
    covariances as the reference-semantics mode.  Only a ring that had more than 32 free slots when the update started (no prior
    saved) and is lapped all the same throws."""
    exe = os.path.join(PKG, "xk_ring_wrap_example")
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, "during_update", str(bsz)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and expect in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("resident", [0, 1])
def test_dense_apply_update_with_a_resident_covariance(xk, resident):
    """Updater::applyUpdate (updater.cpp:117-141) handed a DENSE h by a subclass that bypasses the device-resident construction:
    it used to throw when the covariance is resident; now the covariance is fetched, updated and
    sent back.  Against oracle/ref_np.apply_update."""
    host = C.CDLL(os.path.join(PKG, "libx_host.so"))
    N, M, m = 5, 2, 7
    n = 15 + 6 * N + 3 * M
    rng = np.random.default_rng(31)
    A = rng.standard_normal((n, n))
    P = np.asfortranarray(A @ A.T * 1e-3 + 1e-2 * np.eye(n))
    H = np.asfortranarray(rng.standard_normal((m, n)))
    res, rd = 1e-2 * rng.standard_normal(m), np.full(m, 4e-6)
    Pout, core = np.zeros((n, n), order="F"), np.zeros(16)
    rc = host.x_host_dense_update(_p(P), C.c_int(N), C.c_int(M), _p(H), C.c_int(m), _p(res), _p(rd), C.c_int(resident), _p(Pout), _p(core))
    assert rc == 0
    Pe, corr = ref_np.apply_update(P, H, res, rd)
    assert rel(Pout, Pe) <= 1e-11
    assert rel(core[0:3], corr[0:3]) <= 1e-9 and rel(core[3:6], corr[3:6]) <= 1e-9      # p, v took the correction
