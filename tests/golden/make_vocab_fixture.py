"""Turns the reference's own vocabulary data files (Vocabulary/*.yaml -- DBoW3 binary containers despite the
extension) into plain arrays: x_multi_agent_amd/data/vocab_<name>.npz (the package owns the data it loads).  Data only: node descriptors, the tree, the
word table.  Run in the build container (needs /root/reference and gcc):
    python tests/golden/make_vocab_fixture.py
The stream layout parsed here is Vocabulary::fromStream (third_party/DBow3/src/Vocabulary.cpp:1374-1410) and
DescManip::fromStream (DescManip.cpp:261-268): i32 k, L, scoring, weighting; per node u32 id, u32 parent, f64
weight, i32 cols, rows, type, `cols` bytes; u32 n_words; per word u32 word id, u32 node id."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_build import build_unpack  # noqa: E402

VOCS = {"visual": "/root/reference/Vocabulary/visual_voc_3_4_dbow3.yaml",
        "thermal": "/root/reference/Vocabulary/thermal_voc_3_4_dbow3_calib.yaml"}


def parse(raw):
    nn, = struct.unpack_from("<I", raw, 0)
    k, L, scoring, weighting = struct.unpack_from("<iiii", raw, 4)
    o = 20
    parent = np.full(nn, -1, np.int32)
    weight = np.zeros(nn)
    desc = None
    children = [[] for _ in range(nn)]
    for _ in range(nn - 1):
        nid, par = struct.unpack_from("<II", raw, o)
        w, = struct.unpack_from("<d", raw, o + 8)
        cols, rows, typ = struct.unpack_from("<iii", raw, o + 16)
        assert rows == 1 and typ == 0, "binary (CV_8U) one-row descriptors expected"
        if desc is None:
            desc = np.zeros((nn, cols), np.uint8)
        desc[nid] = np.frombuffer(raw, np.uint8, cols, o + 28)
        parent[nid], weight[nid] = par, w
        children[par].append(nid)            # insertion order = file order (Vocabulary.cpp:1394)
        o += 28 + cols
    nw, = struct.unpack_from("<I", raw, o)
    o += 4
    word_of_node = np.full(nn, -1, np.int32)
    node_of_word = np.zeros(nw, np.int32)
    for _ in range(nw):
        wid, nid = struct.unpack_from("<II", raw, o)
        word_of_node[nid], node_of_word[wid] = wid, nid
        o += 8
    assert o == len(raw)
    kmax = max(len(c) for c in children)
    ch = np.full((nn, kmax), -1, np.int32)
    for i, c in enumerate(children):
        ch[i, :len(c)] = c
    return dict(k=np.int32(k), L=np.int32(L), scoring=np.int32(scoring), weighting=np.int32(weighting), parent=parent,
                weight=weight, desc=desc, children=ch, word_of_node=word_of_node, node_of_word=node_of_word)


def main():
    exe = build_unpack.build()
    for name, path in VOCS.items():
        with tempfile.NamedTemporaryFile(suffix=".raw") as t:
            subprocess.check_call([exe, path, t.name])
            v = parse(open(t.name, "rb").read())
        out = os.path.join(HERE, "..", "..", "x_multi_agent_amd", "data", f"vocab_{name}.npz")
        np.savez_compressed(out, **v)
        print(out, "k", int(v["k"]), "L", int(v["L"]), "nodes", len(v["parent"]), "words", len(v["node_of_word"]),
              "desc bytes", v["desc"].shape[1])


if __name__ == "__main__":
    main()
