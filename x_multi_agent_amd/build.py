"""Builds libxk.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

    python -m x_multi_agent_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "xk_api.hip")
OUT = os.path.join(HERE, "libxk.so")
# every file xk_api.hip includes: a stale libxk.so after editing any of them would silently test old kernels
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))
              if f.endswith((".hip", ".h"))) + [os.path.join(HERE, "..", "include", "xk.h")]   # (xk_fleet.cpp: build_fleet)


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


def build_lab(force=False, verbose=True):
    """lab/libxk.so: the same sources with -DXK_LAB -- environment switches, test hooks (xk_set_option "caqr_poison" ...), debug
    exports and probe kernels (include/xk_lab.h).  Same file name in a directory of its own, so that the C++ examples pick it up
    through LD_LIBRARY_PATH and Python through engine.lib(lab=True); the release library has none of it."""
    out = os.path.join(HERE, "lab", "libxk.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = DEPS + [os.path.join(HERE, "..", "include", "xk_lab.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
           "-DXK_LAB", "-o", out, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_strict(verbose=True):
    """libxk_strict.so: the same sources with -DXK_SYNC_STRICT=1 -- every hand-off of the single-launch kernels as an agent-scope
    release / acquire pair (xk_xcd_sync.hip.h).  2.3x slower; it is the reference the default build's hand-offs are checked
    against (tests/test_gpu_strict_sync.py), selected with XK_LIB_PATH."""
    out = os.path.join(HERE, "libxk_strict.so")
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
           "-DXK_SYNC_STRICT=1", "-o", out, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_fleet(verbose=True):
    """libxk_fleet.so: the RCCL exchange of the CI step (include/xk_fleet.h), host code over libxk.so + librccl.so."""
    out = os.path.join(HERE, "libxk_fleet.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-o", out, os.path.join(HERE, "csrc", "xk_fleet.cpp"),
           "-L" + HERE, "-lxk", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_host(force=False, verbose=True):
    """Host-side C++ mirror of the reference API (host/) and its example driver."""
    root = os.path.join(HERE, "..")
    lib = os.path.join(HERE, "libx_host.so")
    exe = os.path.join(HERE, "xk_host_example")
    srcs = [os.path.join(root, "host", "src", f) for f in sorted(os.listdir(os.path.join(root, "host", "src"))) if f.endswith(".cpp")]
    inc = ["-I" + os.path.join(root, "host", "include"), "-I" + os.path.join(root, "include")]
    link = ["-L" + HERE, "-lxk", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    cmds = [["g++", "-std=c++17", "-O3", "-Wall", "-fPIC", "-shared"] + inc + srcs + ["-o", lib] + link]
    # every host/examples/<name>_main.cpp -> x_multi_agent_amd/xk_<short>_example
    short = {"visual_update": "host", "state_manage": "manage", "place_recognition": "place"}
    for f in sorted(os.listdir(os.path.join(root, "host", "examples"))):
        if not f.endswith("_main.cpp"):
            continue
        name = f[:-len("_main.cpp")]
        out = os.path.join(HERE, "xk_%s_example" % short.get(name, name))
        extra_inc, extra_lib = [], []
        if name == "fleet":   # the one example that talks to the HIP runtime and RCCL itself (device buffers, xk_fleet.h)
            extra_inc = ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-Wno-unused-result"]
            extra_lib = ["-lxk_fleet", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"]
        cmds.append(["g++", "-std=c++17", "-O2", "-Wall"] + inc + extra_inc + [os.path.join(root, "host", "examples", f), "-o", out,
                                                                               "-L" + HERE, "-lx_host"] + extra_lib + link[1:])
    for c in cmds:
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c)
    return lib, exe


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_lab(force="--force" in sys.argv)
    build_fleet()
    build_host()
