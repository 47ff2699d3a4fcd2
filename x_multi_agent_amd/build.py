"""Builds libxk.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

    python -m x_multi_agent_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "xk_api.hip")
OUT = os.path.join(HERE, "libxk.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("xk_api.hip", "xk_feature.hip.h", "xk_linalg.hip.h", "xk_ci.hip.h",
                                               "xk_chi2_table.h")] + [os.path.join(HERE, "..", "include", "xk.h")]


def build(force=False, verbose=True):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
           "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
