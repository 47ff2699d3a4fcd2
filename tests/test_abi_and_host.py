"""The C-ABI library loads and exports every symbol include/xk.h declares; host-side
logic that needs no GPU (argument validation happens before any device call)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from x_multi_agent_amd import engine, synth

ROOT = os.path.join(os.path.dirname(__file__), "..")


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "xk.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:const\s+)?(?:int|long|void|char)\s*\*?\s*(xk_[a-z_A-Z0-9]+)\s*\(", hdr,
                                     flags=re.M)))
    assert len(declared) >= 24
    L = engine.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(engine.SYMBOLS) == declared


def test_release_library_carries_no_lab_exports_and_the_lab_build_carries_both():
    """the shipped library is not the lab.  include/xk_lab.h declares what only the -DXK_LAB build
    (x_multi_agent_amd/lab/libxk.so) exports -- test hooks, debug stamps, probe kernels; the release library has none of it, does not
    reference getenv, and refuses the test-hook options."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "xk_lab.h")).read()
    lab_decl = sorted(set(re.findall(r"^\s*(?:const\s+)?(?:int|long|void|char)\s*\*?\s*(xk_[a-z_A-Z0-9]+)\s*\(", hdr, flags=re.M)))
    assert lab_decl == sorted(engine.LAB_SYMBOLS)
    rel, lab = engine.lib(), engine.lib(lab=True)
    assert not [s for s in lab_decl if hasattr(rel, s)]
    assert not [s for s in lab_decl + engine.SYMBOLS if not hasattr(lab, s)]
    assert lab.xk_is_lab() == 1
    und = lambda p: subprocess.run(["nm", "-D", "--undefined-only", p], capture_output=True, text=True).stdout
    assert "getenv" not in und(engine.LIB_PATH) and "getenv" in und(engine.LAB_LIB_PATH)
    syms = subprocess.run(["nm", "-D", "--defined-only", engine.LIB_PATH], capture_output=True, text=True).stdout
    assert "xk_probe" not in syms and "xk_debug" not in syms


def test_version_strerror_payload_size():
    L = engine.lib()
    assert L.xk_version() >= 100
    assert L.xk_strerror(0) == b"ok" and b"invalid" in L.xk_strerror(1)
    # SimpleState sizes 306 008 B (N=30, M=0) and 955 408 B (M=50)
    # are the reference's payload; ours adds hdr[8] and pads anchors to doubles
    n = 15 + 6 * 30
    assert L.xk_payload_doubles(30, 0) == 8 + 16 + 90 + 120 + n * n
    n = 15 + 6 * 30 + 150
    assert L.xk_payload_doubles(30, 50) == 8 + 16 + 90 + 120 + 150 + 50 + n * n


def test_null_and_bad_arguments_are_status_codes_not_crashes():
    L = engine.lib()
    assert L.xk_create(0, 1, 0, 10, None) == 1          # XK_EINVAL: out == NULL
    h = C.c_void_p()
    assert L.xk_create(0, 100, 0, 10, C.byref(h)) == 1   # n_poses_max > 64
    assert L.xk_destroy(None) == 0
    assert L.xk_stage_window(None, None, None, 3) == 1
    assert L.xk_run_steps(None, C.c_double(1e-3), 1) == 1


def test_synth_is_deterministic_and_platform_independent():
    r = synth.SplitMix(0x5EED0000)
    u = r.u64(3)
    # splitmix64 of (seed + i*golden), fixed forever
    assert [int(x) for x in u] == [int(x) for x in synth.SplitMix(0x5EED0000).u64(3)]
    a, b = synth.make_config(1), synth.make_config(1)
    for k in ("C_q_G", "G_p_C", "obs_xy", "P"):
        assert np.array_equal(a[k], b[k])
    c = synth.make_config(1, agent_id=1)
    assert not np.array_equal(a["obs_xy"], c["obs_xy"])
    P = a["P"]
    assert np.abs(P - P.T).max() == 0 and np.linalg.eigvalsh(P).min() > 0
    assert a["trk_off"][0] == 0 and a["trk_off"][-1] == a["obs_xy"].shape[0]


def test_splitmix_reference_values():
    # splitmix64 with seed 0: first outputs of the canonical generator
    r = synth.SplitMix(0)
    assert [hex(int(x)) for x in r.u64(3)] == ["0xe220a8397b1dcdaf", "0x6e789e6aa1b965f4", "0x6c45d188009454f"]


def test_fleet_library_exports_its_header():
    """libxk_fleet.so (RCCL exchange of the CI step) loads next to libxk.so and exports everything xk_fleet.h declares."""
    hdr = open(os.path.join(ROOT, "include", "xk_fleet.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:const\s+)?(?:int|long|void|char)\s*\*?\s*(xk_fleet_[a-z_A-Z0-9]+)\s*\(", hdr, flags=re.M)))
    assert len(declared) == 9
    engine.lib()
    path = os.path.join(ROOT, "x_multi_agent_amd", "libxk_fleet.so")
    if not os.path.exists(path):
        from x_multi_agent_amd import build
        build.build_fleet(verbose=False)
    F = C.CDLL(path)
    assert not [s for s in declared if not hasattr(F, s)]
    assert F.xk_fleet_create(None, None, 1, 0, None) == 1 and F.xk_fleet_all_gather(None, None, None, 0) == 1    # XK_EINVAL
