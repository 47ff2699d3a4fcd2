"""Extended-precision (x87 long double, eps = 1.08e-19) restatement of the numerically delicate half of the path --
chi-square gate, QR compression, Kalman update (msckf_update.cpp:457-463, vio_updater.cpp:487-512, updater.cpp:117-141)
-- used as a THIRD opinion between the two double-precision restatements (oracle/ref_np.py, oracle/xk_oracle.c): the
reference has no golden vectors and cannot be compiled here (parity unpinned), so the restatements are
cross-examined instead.  Test infrastructure only (tests/test_oracle_precision.py).

No LAPACK exists for this type: Householder QR, Cholesky and the triangular solves are written out; numpy only supplies
the element-wise arithmetic and matmul loops in long double."""
import numpy as np

LD = np.longdouble


def householder_r(A):
    """R factor (upper trapezoid, min(m, n) rows) of A by Householder reflections (Eigen::HouseholderQR's algorithm)."""
    A = np.array(A, dtype=LD)
    m, n = A.shape
    for k in range(min(m - 1, n)):
        x = A[k:, k]
        tail = np.dot(x[1:], x[1:])
        if tail == 0:
            continue
        nrm = np.sqrt(x[0] * x[0] + tail)
        beta = -nrm if x[0] >= 0 else nrm
        v = x.copy()
        v[0] = x[0] - beta
        tau = LD(2) / np.dot(v, v)
        A[k:, k:] -= tau * np.outer(v, v @ A[k:, k:])
        A[k + 1:, k] = 0
    return np.triu(A[:min(m, n)])


def cholesky(S):
    S = np.array(S, dtype=LD)
    n = S.shape[0]
    L = np.zeros((n, n), dtype=LD)
    for j in range(n):
        d = S[j, j] - np.dot(L[j, :j], L[j, :j])
        if not d > 0:
            raise np.linalg.LinAlgError("not positive definite")
        L[j, j] = np.sqrt(d)
        if j + 1 < n:
            L[j + 1:, j] = (S[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def solve_lower(L, B):
    B = np.array(B, dtype=LD)
    X = np.zeros_like(B)
    for i in range(L.shape[0]):
        X[i] = (B[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X


def gate_gamma(jac0, res0, P, var_img):
    """gamma = res0^T (jac0 P jac0^T + var I)^-1 res0  (msckf_update.cpp:457-461)."""
    J, r, Pl = np.array(jac0, dtype=LD), np.array(res0, dtype=LD), np.array(P, dtype=LD)
    S = J @ Pl @ J.T + LD(var_img) * np.eye(J.shape[0], dtype=LD)
    y = solve_lower(cholesky(S), r)
    return np.dot(y, y)


def compress_and_update(h, res, P, sigma_img):
    """applyQRDecomposition (vio_updater.cpp:487-512) + applyUpdate (updater.cpp:117-141, correction_total = 0)."""
    h, res, P = np.array(h, dtype=LD), np.array(res, dtype=LD), np.array(P, dtype=LD)
    rows, cols = h.shape
    var = LD(sigma_img) ** 2
    if rows > cols + 1:
        R = householder_r(np.hstack([h, res.reshape(-1, 1)]))
        h, res = R[:cols, :cols], R[:cols, cols]
    n = P.shape[0]
    HP = h @ P
    S = HP @ h.T + var * np.eye(h.shape[0], dtype=LD)
    L = cholesky(S)
    X = solve_lower(L, HP)                      # L^-1 H P
    y = solve_lower(L, res)
    Pn = P - X.T @ X                            # (I - K H) P with K = P H^T S^-1
    Pn = (Pn + Pn.T) / 2
    corr = X.T @ y
    return Pn, corr
