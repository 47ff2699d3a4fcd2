"""The C++ mirror of x::Database / VLAD / Keyframe and of findCorrespondences' matching half
(host/include/x/place_recognition/database.h) driven through its example binary, against oracle/ref_pr.py."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import ref_pr
from x_multi_agent_amd import place, synth

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd")


def _ints(a):
    return " ".join(str(int(x)) for x in np.asarray(a).ravel())


@pytest.mark.parametrize("vname", ["visual", "pruned"])
def test_cpp_database_and_matching(tmp_path, vname):
    exe = os.path.join(PKG, "xk_place_example")
    if not os.path.exists(exe):
        from x_multi_agent_amd import build
        build.build_host()
    v = place.load_vocabulary("visual") if vname == "visual" else synth.make_vocabulary(6, 2, 32, seed=8, prune=0.3)
    voc = ref_pr.Vocabulary(v)
    thr, min_d, ratio = 0.6, 60.0, 0.8
    rng = np.random.default_rng(3)
    scenes = [synth.make_descriptors(70, 32, seed=40 + s) for s in range(3)]
    lines = [f"{int(v['k'])} {int(v['L'])} {v['desc'].shape[0]} {v['children'].shape[1]} 32 {len(v['node_of_word'])} {thr} {min_d} {ratio}",
             _ints(v["desc"]), _ints(v["children"]), _ints(v["word_of_node"]), _ints(v["node_of_word"])]
    ora = ref_pr.Database(voc, thr)
    expect = []
    for i in range(18):
        d = synth.observe_descriptors(scenes[i % 3], int(rng.integers(0, 10)), seed=i)
        lines.append(f"A {i} {len(d)} {_ints(d)}")
        ora.add_keyframe(ref_pr.Keyframe(d, tag=i))
        expect.append(f"A {len(ora.keyframes)}")
        for uav in (2, 2, 5):
            q = synth.observe_descriptors(scenes[int(rng.integers(0, 3))], int(rng.integers(0, 25)), seed=500 + 3 * i + uav)
            lines.append(f"F {uav} {len(q)} {_ints(q)}")
            kf, idx, sc = ora.find_candidate(uav, ref_pr.compute_vlad(voc, q))
            bits = struct.unpack("<Q", struct.pack("<d", sc))[0]
            expect.append(f"F {idx} {bits:x} {kf.tag if kf is not None else -1} {len(kf.descriptors) if kf is not None else 0}")
    for nq, nt in ((40, 70), (70, 70), (3, 1)):
        t = synth.observe_descriptors(scenes[0][:nt], 5, seed=70 + nq)
        qd = synth.observe_descriptors(scenes[0][rng.integers(0, max(nt, 1), nq)], 8, seed=80 + nq)   # repeats -> duplicate claims
        ncm, ncs, nrm, nrs = nt // 3, nt // 3, nq // 3, nq // 3
        lines.append(f"M {nq} {nt} {ncm} {ncs} {nrm} {nrs} {_ints(qd)} {_ints(t)}")
        good = ref_pr.good_matches(*ref_pr.knn2(qd, t), min_d, ratio)
        kinds = {"msckf": 0, "slam": 1, "opp_slam": 2, "opp_opp": 3}
        cls = ref_pr.classify(good, ncm, ncs, nrm, nrs)
        expect.append(f"M {len(good)}" + "".join(f" {q}:{tt}" for q, tt in good) + " |" +
                      "".join(f" {kinds[k]}:{c}:{r}" for k, c, r in cls))
    fin = tmp_path / "case.txt"
    fin.write_text("\n".join(lines) + "\n")
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(fin)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = r.stdout.strip().splitlines()
    assert len(got) == len(expect)
    for g, e in zip(got, expect):
        assert g == e
