"""IMU propagation closed forms (host side): the restatements of Propagator::discreteStateTransition /
quaternionIntegrator / propagateState in oracle/ref_np.py and in the C++ mirror (host/src/propagator.cpp), pinned by
first principles (matrix exponential of the continuous error dynamics, exact quaternion kinematics, semigroup property)
and against each other; and what is known about the reference's machine-generated q_d from its OWN outputs
(tests/golden/propagator_qd.npz, written by tests/golden/make_propagator_golden.py from the reference's statements)."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.linalg import expm

from helpers import GOLDEN_DIR, rel
from oracle import ref_np

PKG = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd")
c_dp = C.POINTER(C.c_double)


def _p(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(c_dp)


@pytest.fixture(scope="module")
def host():
    from x_multi_agent_amd import engine
    engine.lib()                                   # libx_host.so links libxk.so; loading needs no GPU
    path = os.path.join(PKG, "libx_host.so")
    if not os.path.exists(path):
        from x_multi_agent_amd import build
        build.build_host(verbose=False)
    return C.CDLL(path)


def _cases(n=12, seed=5):
    rng = np.random.default_rng(seed)
    for i in range(n):
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        yield (float(rng.choice([0.0025, 0.005, 0.01, 0.02])), rng.standard_normal(3) * (0.1 if i % 2 else 1.0),
               rng.standard_normal(3) * 2 + np.array([0, 0, 9.81]), q)


def _continuous_F(e_w, e_a, q):
    """Error-state dynamics the closed form integrates (Weiss 2012 eq. 2.24): d(dp)=dv, d(dv)=-C[a]x dth - C dba,
    d(dth) = -[w]x dth - dbw."""
    Cq = ref_np.quat_to_rot(q)
    F = np.zeros((15, 15))
    F[0:3, 3:6] = np.eye(3)
    F[3:6, 6:9] = -Cq @ ref_np.skew(e_a)
    F[3:6, 12:15] = -Cq
    F[6:9, 6:9] = -ref_np.skew(e_w)
    F[6:9, 9:12] = -np.eye(3)
    return F


def test_state_transition_is_the_truncated_matrix_exponential():
    for dt, e_w, e_a, q in _cases():
        Fd = ref_np.discrete_state_transition(dt, e_w, e_a, q)
        Fe = expm(_continuous_F(e_w, e_a, q) * dt)
        # blocks are series truncated two or three orders above their leading term
        for (r, c, lead) in [((0, 3), (3, 6), 1), ((0, 3), (6, 9), 2), ((0, 3), (9, 12), 3), ((0, 3), (12, 15), 2),
                             ((3, 6), (6, 9), 1), ((3, 6), (9, 12), 2), ((3, 6), (12, 15), 1), ((6, 9), (6, 9), 0), ((6, 9), (9, 12), 1)]:
            a, b = Fd[r[0]:r[1], c[0]:c[1]], Fe[r[0]:r[1], c[0]:c[1]]
            scale = (np.linalg.norm(e_w) + 1) ** 3 * (np.linalg.norm(e_a) + 1) * dt ** (lead + 3)
            assert np.abs(a - b).max() <= 2.0 * scale, (r, c)
        # everything the closed form leaves at identity / zero really is
        mask = np.ones((15, 15), bool)
        for r, c in [((0, 3), (3, 15)), ((3, 6), (6, 15)), ((6, 9), (6, 12))]:
            mask[r[0]:r[1], c[0]:c[1]] = False
        assert np.array_equal(Fd[mask], np.eye(15)[mask])


def test_state_transition_semigroup():
    for dt, e_w, e_a, q in _cases(6, seed=9):
        F2 = ref_np.discrete_state_transition(2 * dt, e_w, e_a, q)
        F1 = ref_np.discrete_state_transition(dt, e_w, e_a, q)
        assert np.abs(F2 - F1 @ F1).max() <= 50 * (np.linalg.norm(e_w) + 1) ** 3 * (np.linalg.norm(e_a) + 1) * dt ** 3


def test_quaternion_integrator_matches_exact_kinematics():
    rng = np.random.default_rng(3)
    for _ in range(8):
        w, dt = rng.standard_normal(3), 0.005
        exact = expm(ref_np.omega_matrix(w) * 0.5 * dt)             # constant rate: q(t+dt) = exp(Omega dt / 2) q(t)
        got = ref_np.quaternion_integrator(w, w, dt)
        assert np.abs(got - exact).max() <= (np.linalg.norm(w) * dt) ** 5
        q0 = rng.standard_normal(4); q0 /= np.linalg.norm(q0)
        q1 = got @ q0
        assert abs(np.linalg.norm(q1) - 1) <= 1e-12


def test_propagate_state_constant_acceleration():
    g = np.array([0, 0, -9.81])
    q = np.array([0.1, -0.2, 0.05, 0.97]); q /= np.linalg.norm(q)
    a_world = np.array([0.3, -0.1, 0.2])
    a_m = ref_np.quat_to_rot(q).T @ (a_world - g)                    # specific force for that acceleration, no rotation
    s0 = dict(time=1.0, p=np.array([1.0, 2, 3]), v=np.array([0.1, 0.2, 0.3]), q=q, b_w=np.zeros(3), b_a=np.zeros(3),
              w_m=np.zeros(3), a_m=a_m)
    s1 = dict(time=1.01, w_m=np.zeros(3), a_m=a_m)
    ref_np.propagate_state(s0, s1, g)
    assert rel(s1["v"], s0["v"] + a_world * 0.01) <= 1e-13
    assert rel(s1["p"], s0["p"] + s0["v"] * 0.01 + 0.5 * a_world * 1e-4) <= 1e-13
    assert rel(s1["q"], q) <= 1e-15


def test_host_mirror_closed_forms_match_the_oracle(host):
    rng = np.random.default_rng(11)
    for dt, e_w, e_a, q in _cases(8, seed=21):
        out = np.zeros(225)
        host.x_host_discrete_state_transition(C.c_double(dt), _p(e_w), _p(e_a), _p(q), _p(out))
        assert np.abs(out.reshape(15, 15, order="F") - ref_np.discrete_state_transition(dt, e_w, e_a, q)).max() <= 1e-15
        nz = [float(10 ** rng.uniform(-4, -1)) for _ in range(4)]
        host.x_host_process_noise_model(C.c_double(dt), _p(q), _p(e_w), _p(e_a), *[C.c_double(v) for v in nz], _p(out))
        Qm = ref_np.process_noise_model(dt, e_w, e_a, q, *nz)
        assert rel(out.reshape(15, 15, order="F"), Qm) <= 1e-12
        w1 = e_w + 0.1 * rng.standard_normal(3)
        o16 = np.zeros(16)
        host.x_host_quaternion_integrator(_p(e_w), _p(w1), C.c_double(dt), _p(o16))
        assert np.abs(o16.reshape(4, 4) - ref_np.quaternion_integrator(e_w, w1, dt)).max() <= 1e-15
        # one IMU step of the state
        s0 = dict(time=2.0, p=rng.standard_normal(3), v=rng.standard_normal(3), q=q, b_w=0.01 * rng.standard_normal(3),
                  b_a=0.05 * rng.standard_normal(3), w_m=e_w, a_m=e_a)
        s1 = dict(time=2.0 + dt, w_m=w1, a_m=e_a + 0.2 * rng.standard_normal(3))
        g = np.array([0, 0, -9.81])
        flat = lambda s: np.concatenate([[s["time"]], s.get("p", np.zeros(3)), s.get("v", np.zeros(3)), s.get("q", np.array([0, 0, 0, 1.0])),
                                         s.get("b_w", np.zeros(3)), s.get("b_a", np.zeros(3)), s["w_m"], s["a_m"]])
        f0, f1 = flat(s0), flat(s1)
        host.x_host_propagate_state(_p(f0), f1.ctypes.data_as(c_dp), _p(g))
        ref_np.propagate_state(s0, s1, g)
        assert rel(f1[1:4], s1["p"]) <= 1e-14 and rel(f1[4:7], s1["v"]) <= 1e-14 and rel(f1[7:11], s1["q"]) <= 1e-14


def test_process_noise_model_properties():
    for dt, e_w, e_a, q in _cases(4, seed=2):
        Q = ref_np.process_noise_model(dt, e_w, e_a, q, 1e-3, 1e-5, 1e-2, 1e-4)
        assert np.array_equal(Q, Q.T) and np.linalg.eigvalsh(Q).min() >= -1e-22
        assert rel(Q[3:6, 3:6], 1e-4 * dt * np.eye(3)) <= 1e-2            # accelerometer noise dominates the velocity block
        assert rel(Q[9:12, 9:12], 1e-10 * dt * np.eye(3)) <= 1e-14 and rel(Q[12:15, 12:15], 1e-8 * dt * np.eye(3)) <= 1e-14


def test_reference_qd_fixture_facts():
    """What the reference's own q_d outputs show (the reason the mirror does not restate it): bias blocks are n^2 dt I, the
    matrix is NOT symmetric, 78 entries are never assigned, and it is not the integral the model computes."""
    g = np.load(os.path.join(GOLDEN_DIR, "propagator_qd.npz"))
    worst_model, asym = 0.0, 0.0
    for i in range(len(g["dt"])):
        Q, dt = g["Q"][i], float(g["dt"][i])
        n_w, n_bw, n_a, n_ba = g["noise"][i]
        assert rel(np.diag(Q)[9:12], n_bw ** 2 * dt * np.ones(3)) <= 1e-14
        assert rel(np.diag(Q)[12:15], n_ba ** 2 * dt * np.ones(3)) <= 1e-14
        assert np.count_nonzero(Q) <= 147
        asym = max(asym, np.abs(Q - Q.T).max() / np.abs(Q).max())
        worst_model = max(worst_model, rel(ref_np.process_noise_model(dt, g["e_w"][i], g["e_a"][i], g["q"][i], n_w, n_bw, n_a, n_ba), Q))
        # the f_d column of the fixture is the restatement's (regression pin)
        assert np.array_equal(g["F"][i], ref_np.discrete_state_transition(dt, g["e_w"][i], g["e_a"][i], g["q"][i]))
    assert asym > 1e-3 and worst_model > 0.1


def test_reference_qd_goes_in_through_the_hook(host, capfd):
    """A drop-in hands the reference's own discreteProcessNoiseCov to the mirror (Propagator::setProcessNoiseFunction): the
    transition of an IMU step then carries exactly those numbers (here: the fixture generated from the reference's
    statements), next to the restated f_d; without the hook the default model says on stderr that it is NOT the reference's."""
    g = np.load(os.path.join(GOLDEN_DIR, "propagator_qd.npz"))
    for i in (0, 7, 23):
        dt = float(g["dt"][i])
        s0 = np.zeros(23); s1 = np.zeros(23)
        s0[0], s1[0] = 3.0, 3.0 + dt
        s0[7:11] = s1[7:11] = g["q"][i]
        s1[17:20], s1[20:23] = g["e_w"][i], g["e_a"][i]
        Qd = np.asfortranarray(g["Q"][i])
        fd, qd = np.zeros((15, 15), order="F"), np.zeros((15, 15), order="F")
        calls = host.x_host_transition(_p(s0), _p(s1), _p(Qd.ravel(order="F")), C.c_int(1), fd.ctypes.data_as(c_dp), qd.ctypes.data_as(c_dp))
        assert calls == 1 and np.array_equal(qd, g["Q"][i]) and rel(fd, g["F"][i]) <= 1e-14
    # without the hook: the clean model, announced once
    calls = host.x_host_transition(_p(s0), _p(s1), _p(Qd.ravel(order="F")), C.c_int(0), fd.ctypes.data_as(c_dp), qd.ctypes.data_as(c_dp))
    n_w, n_bw, n_a, n_ba = 0.0013, 0.00013, 0.0083, 0.00083        # ImuNoise defaults of the mirror
    assert calls == 0 and rel(qd, ref_np.process_noise_model(dt, g["e_w"][23], g["e_a"][23], g["q"][23], n_w, n_bw, n_a, n_ba)) <= 1e-12
    assert not np.array_equal(qd, g["Q"][23])


def test_simple_state_payload_bridge(host):
    from x_multi_agent_amd import fleet
    rng = np.random.default_rng(4)
    N, M = 6, 3
    n = 15 + 6 * N + 3 * M
    A = rng.standard_normal((n, n))
    q = rng.standard_normal((N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    pay = fleet.pack_payload_host(3, 12.5, rng.standard_normal(16), q, rng.standard_normal((N, 3)), rng.standard_normal(3 * M),
                                  np.array([0, 2, 5]), A @ A.T, N, M)
    out, lists = np.zeros_like(pay), np.zeros(7 * N)
    assert pay.size == fleet.payload_layout(N, M)["total"]
    assert host.x_host_simple_state_roundtrip(_p(pay), C.c_int(N), C.c_int(M), _p(out), _p(lists)) == 0
    assert np.array_equal(out, pay)
    u = fleet.unpack_payload(pay, N, M)
    assert np.array_equal(lists[:4 * N].reshape(N, 4), u["C_q_G"]) and np.array_equal(lists[4 * N:].reshape(N, 3), u["G_p_C"])
    bad = pay.copy(); bad[2] = N + 1
    assert host.x_host_simple_state_roundtrip(_p(bad), C.c_int(N), C.c_int(M), _p(out), _p(lists)) == 1


def test_inverse_depths_of_new_standard_slam_features(host):
    obs = np.array([[0.1, -0.2], [0.3, 0.05], [-0.4, 0.25]])
    out = np.zeros(9)
    host.x_host_inverse_depths_new(_p(obs), C.c_int(3), C.c_double(0.5), _p(out))
    assert np.array_equal(out.reshape(3, 3), np.column_stack([obs, np.full(3, 0.5)]))    # slam_update.cpp:229-242
