"""Golden vectors for the IMU covariance-propagation closed forms, evaluated FROM THE REFERENCE ITSELF.

The reference cannot be compiled here (Eigen), but Propagator::discreteProcessNoiseCov (src/x/ekf/propagator.cpp:207-840)
is 630 lines of scalar arithmetic on doubles (`const double tNNN = ...;` / `q_d(i, j) = ...;`).  This script reads those
statements from /root/reference AT GENERATION TIME, evaluates them with Python floats (IEEE double, same operation
order: the expressions are fully parenthesised sums and products) for seeded inputs, and stores inputs and outputs.
Nothing of the reference's text is kept: the fixture is numbers.  discreteStateTransition (:110-164) is Eigen 3x3
algebra, not scalar text, so its fixture entries come from the restatement and are pinned by the finite-difference /
semigroup checks in tests/test_oracle_propagator.py instead.

    python tests/golden/make_propagator_golden.py      (needs /root/reference; run in the build container)
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
SRC = "/root/reference/src/x/ekf/propagator.cpp"

IDX = dict(kIdxP=0, kIdxV=3, kIdxQ=6, kIdxBw=9, kIdxBa=12)


def reference_statements():
    text = open(SRC).read()
    a = text.index("CoreCovMatrix Propagator::discreteProcessNoiseCov(")
    a = text.index("{", a) + 1
    b = text.index("return q_d;", a)
    body = re.sub(r"//[^\n]*", "", text[a:b])
    return [s.strip().replace("\n", " ") for s in body.split(";") if s.strip()]


def split_top_level_commas(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def to_python(expr):
    expr = re.sub(r"\bq\.([wxyz])\(\)", r"q_\1", expr)
    expr = re.sub(r"\b(e_w|e_a)\((\d)\)", r"\1[\2]", expr)
    return expr


def evaluate_reference_qd(stmts, dt, q_xyzw, e_w, e_a, n_w, n_bw, n_a, n_ba):
    env = dict(IDX, dt=dt, q_x=q_xyzw[0], q_y=q_xyzw[1], q_z=q_xyzw[2], q_w=q_xyzw[3], e_w=list(e_w), e_a=list(e_a),
               n_w=n_w, n_bw=n_bw, n_a=n_a, n_ba=n_ba)
    Q = np.zeros((15, 15))
    for s in stmts:
        if s.startswith("const double"):
            for decl in split_top_level_commas(s[len("const double"):]):
                name, expr = decl.split("=", 1)
                env[name.strip()] = eval(to_python(expr), {}, env)
        elif s.startswith("CoreCovMatrix q_d"):
            continue
        elif s.startswith("q_d("):
            m = re.match(r"q_d\((.*?),(.*?)\)\s*=(.*)$", s)
            i, j = eval(m.group(1), {}, env), eval(m.group(2), {}, env)
            Q[i, j] = eval(to_python(m.group(3)), {}, env)
        else:
            raise RuntimeError("unexpected statement: " + s[:60])
    return Q


def main():
    from oracle import ref_np
    stmts = reference_statements()
    rng = np.random.default_rng(20240928)
    cases = []
    for i in range(24):
        dt = float(rng.choice([0.0025, 0.005, 0.01, 0.02, 0.05]))
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        e_w = rng.standard_normal(3) * (0.05 if i % 3 else 1.5)      # hover-like and aggressive rates
        e_a = rng.standard_normal(3) * 2.0 + np.array([0, 0, 9.81])
        n = dict(n_w=float(10 ** rng.uniform(-4, -2)), n_bw=float(10 ** rng.uniform(-6, -3)),
                 n_a=float(10 ** rng.uniform(-3, -1)), n_ba=float(10 ** rng.uniform(-5, -2)))
        Q = evaluate_reference_qd(stmts, dt, q, e_w, e_a, **n)
        F = ref_np.discrete_state_transition(dt, e_w, e_a, q)
        cases.append(dict(dt=dt, q=q, e_w=e_w, e_a=e_a, noise=np.array([n["n_w"], n["n_bw"], n["n_a"], n["n_ba"]]), Q=Q, F=F))
    out = os.path.join(HERE, "propagator_qd.npz")
    np.savez_compressed(out, **{k: np.array([c[k] for c in cases]) for k in cases[0]},
                        source=np.array("q_d: evaluated from the reference's own statements (propagator.cpp:207-840); F: restatement"))
    print("wrote", out, "max |Q| =", max(np.abs(c["Q"]).max() for c in cases),
          "max asymmetry =", max(np.abs(c["Q"] - c["Q"].T).max() for c in cases))


if __name__ == "__main__":
    main()
