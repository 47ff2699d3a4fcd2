"""Mathematical invariants of the path, checked on the
CPU restatements -- these stand in for the fixtures the reference does not have."""
import numpy as np
import pytest

from helpers import rel
from oracle import ref_np
from x_multi_agent_amd import synth

# known answers: scipy.stats.chi2.ppf == boost::math::quantile(chi_squared)
CHI2 = [(0.95, 1, 3.841458820694124), (0.95, 3, 7.814727903251179), (0.95, 7, 14.067140449340169),
        (0.95, 17, 27.58711163827534), (0.95, 27, 40.113272069413625), (0.95, 57, 75.62374846937608),
        (0.95, 97, 120.98964369660958), (0.9, 1, 2.705543454095404), (0.9, 3, 6.251388631170325),
        (0.9, 4, 7.779440339734858), (0.9, 20, 28.41198058430563), (0.9, 60, 74.3970057193686)]


@pytest.mark.parametrize("p,dof,val", CHI2)
def test_chi2_known_answers(oracle_c, p, dof, val):
    assert abs(oracle_c.chi2inv(p, dof) - val) <= 2e-12 * val
    assert abs(ref_np.chi2inv(p, dof) - val) <= 1e-13 * val


def test_chi2_table_of_the_product_matches_independent_quantile(oracle_c):
    """The product ships a table (csrc/xk_chi2_table.h); the oracle computes quantiles numerically."""
    import os
    import re
    path = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd", "csrc", "xk_chi2_table.h")
    txt = open(path).read()
    for name, p in (("XK_CHI2_095", 0.95), ("XK_CHI2_090", 0.90)):
        body = txt.split(name + "[XK_CHI2_LEN] = {")[1].split("};")[0]
        vals = [float(v) for v in re.findall(r"[-+0-9.eE]+", body)]
        assert len(vals) == 513 and vals[0] == 0.0
        for dof in list(range(1, 130)) + [200, 300, 512]:
            assert abs(vals[dof] - oracle_c.chi2inv(p, dof)) <= 5e-12 * vals[dof], (name, dof)


@pytest.fixture(scope="module")
def case():
    sc = synth.make_scenario(8, 24, 0, seed=123)
    return sc


def _one_track(sc, k=0):
    tr = synth.tracks_as_list(sc)[k]
    L = len(tr)
    q, p = sc["C_q_G"][-L:], sc["G_p_C"][-L:]
    ivd, _ = ref_np.triangulate_gn(q, p, tr)
    gpf = ref_np.global_feature_position(ivd, q[-1], p[-1])
    return tr, gpf


def test_nullspace_basis_properties(case):
    tr, gpf = _one_track(case)
    jac, hf, res = ref_np.msckf_track_jacobians(tr, case["C_q_G"], case["G_p_C"], case["n_poses_max"], gpf,
                                                case["P"].shape[0])
    a_up, a = ref_np.left_nullspace(hf)
    assert np.abs(a.T @ hf).max() <= 1e-12
    assert np.abs(a.T @ a - np.eye(a.shape[1])).max() <= 1e-13
    # gamma is invariant to the choice of basis: rotate A by a random orthogonal matrix
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((a.shape[1], a.shape[1])))
    P, s2 = case["P"], case["sigma_img"] ** 2

    def gamma(A):
        r0, j0 = A.T @ res, A.T @ jac
        return r0 @ np.linalg.solve(j0 @ P @ j0.T + s2 * np.eye(A.shape[1]), r0)
    assert abs(gamma(a) - gamma(a @ q)) <= 1e-9 * gamma(a)


def test_observability_constraint(case):
    tr, gpf = _one_track(case, 3)
    n = case["P"].shape[0]
    N = case["n_poses_max"]
    jac, hf, res = ref_np.msckf_track_jacobians(tr, case["C_q_G"], case["G_p_C"], N, gpf, n)
    L = len(tr)
    npz = len(case["G_p_C"])
    for i in range(L):
        pos = npz - L + i
        R = ref_np.quat_to_rot(case["C_q_G"][pos])
        Jp = jac[2 * i:2 * i + 2, 15 + 3 * pos:18 + 3 * pos]
        Ja = jac[2 * i:2 * i + 2, 15 + 3 * N + 3 * pos:18 + 3 * N + 3 * pos]
        assert np.abs(Jp @ (R @ ref_np.GRAV)).max() <= 1e-10                       # msckf_update.cpp:398-400
        u = ref_np.skew(gpf - case["G_p_C"][pos]) @ ref_np.GRAV
        assert np.abs(Ja @ u).max() <= 1e-9                                          # :402-406


def test_unconstrained_jacobians_match_finite_differences(case):
    """d z_hat / d(position, small-angle attitude) before the OC projection (msckf_update.cpp:369-388)."""
    tr, gpf = _one_track(case, 5)
    pos = len(case["G_p_C"]) - 1
    q, p = case["C_q_G"][pos], np.array(case["G_p_C"][pos])
    R = ref_np.quat_to_rot(q)

    def proj(Rm, pv):
        c = Rm.T @ (gpf - pv)
        return c[:2] / c[2]
    c = R.T @ (gpf - p)
    Ji = np.array([[1 / c[2], 0, -c[0] / c[2] ** 2], [0, 1 / c[2], -c[1] / c[2] ** 2]])
    Jp, Ja = -Ji @ R.T, Ji @ ref_np.skew(c)
    eps = 1e-6
    for k in range(3):
        e = np.zeros(3)
        e[k] = eps
        fd_p = (proj(R, p + e) - proj(R, p - e)) / (2 * eps)
        assert np.abs(fd_p - Jp[:, k]).max() <= 1e-7
        Rp = R @ ref_np.quat_to_rot(ref_np.error_quat(e))
        Rm = R @ ref_np.quat_to_rot(ref_np.error_quat(-e))
        fd_a = (proj(Rp, p) - proj(Rm, p)) / (2 * eps)
        assert np.abs(fd_a - Ja[:, k]).max() <= 1e-6


def test_compressed_equals_uncompressed_and_row_ops(oracle_c, case):
    jac, res, cov, info = oracle_c.msckf_update(case)
    used = info["rows_used"]
    P, s = case["P"], case["sigma_img"]
    n = P.shape[0]
    T, z, rd, did = oracle_c.qr_compress(jac, res, s)
    assert did
    Pc, cc = oracle_c.apply_update(P, T, z, rd)
    # uncompressed (only the nonzero rows, R = sigma^2 I)
    Hu, ru = jac[:used], res[:used]
    Pu, cu = ref_np.apply_update(P, Hu, ru, np.full(used, s * s))
    assert rel(Pc, Pu) <= 1e-9 and rel(cc, cu) <= 1e-8
    # zero rows with R=1 (Q1), a row permutation and sign flips change nothing
    rng = np.random.default_rng(1)
    perm = rng.permutation(jac.shape[0])
    sg = rng.choice([-1.0, 1.0], size=jac.shape[0])
    T2, z2, rd2, _ = oracle_c.qr_compress(sg[:, None] * jac[perm], sg * res[perm], s)
    P2, c2 = oracle_c.apply_update(P, T2, z2, rd2)
    assert rel(P2, Pc) <= 1e-10 and rel(c2, cc) <= 1e-9
    # posterior: symmetric, PSD, below the prior
    assert np.abs(Pc - Pc.T).max() == 0.0
    assert np.linalg.eigvalsh(Pc).min() >= -1e-12 * np.linalg.norm(Pc)
    assert np.linalg.eigvalsh(P - Pc).min() >= -1e-10 * np.linalg.norm(P)


def test_apply_ci_reduces_to_apply_update(oracle_c, case):
    rng = np.random.default_rng(2)
    P = case["P"]
    n = P.shape[0]
    H = rng.standard_normal((5, n))
    H[:, :15] = 0
    r = rng.standard_normal(5) * 1e-2
    R = np.full(5, 1e-4)
    S = H @ P @ H.T + np.diag(R)
    Pa, ca = oracle_c.apply_ci(P, H, r, S)
    Pu, cu = oracle_c.apply_update(P, H, r, R)
    assert rel(Pa, Pu) <= 1e-12 and rel(ca, cu) <= 1e-12


def test_fuse_ci_k0_and_weights(oracle_c, case):
    rng = np.random.default_rng(3)
    P = case["P"]
    H = rng.standard_normal((3, P.shape[0]))
    S, w = oracle_c.fuse_ci_msckf(P, H, [], [], 0.3)
    assert w == 1.0 and rel(S, H @ P @ H.T) <= 1e-13
    for bad in (0.0, 1.5, -2.0, -0.3):
        with pytest.raises(RuntimeError):
            oracle_c.fuse_ci_slam(P, H, P, H, bad)
    with pytest.raises(RuntimeError):
        ref_np.fuse_ci_slam(P, H, P, H, 0.0)


def test_state_correct(oracle_c):
    rng = np.random.default_rng(4)
    N, M = 4, 2
    st = dict(p=rng.standard_normal(3), v=rng.standard_normal(3), q=np.array([0.1, -0.2, 0.3, 0.9]),
              b_w=rng.standard_normal(3), b_a=rng.standard_normal(3), p_array=rng.standard_normal(3 * N),
              q_array=np.tile(np.array([0.0, 0.0, 0.0, 1.0]), N), f_array=rng.standard_normal(3 * M))
    st["q"] /= np.linalg.norm(st["q"])
    corr = 1e-2 * rng.standard_normal(15 + 6 * N + 3 * M)
    corr[15 + 3 * N:15 + 3 * N + 3] = 0.0  # exact-zero small angle -> identity branch (state.cpp:274)
    a, b = oracle_c.state_correct(st, corr), ref_np.state_correct(st, corr)
    for k in st:
        assert rel(a[k], b[k]) <= 1e-14, k
    assert np.array_equal(a["q_array"][:4], np.array([0.0, 0.0, 0.0, 1.0]))
    assert abs(np.linalg.norm(a["q"]) - 1) <= 1e-15
