"""GPU parity (bit-exact) of the place-recognition request filter against oracle/ref_pr.py, through the C ABI:
VLAD (xk_pr_compute_vlad), keyframe database (xk_pr_add_keyframe / xk_pr_find_candidate / xk_pr_keyframe) and
2-NN descriptor matching (xk_pr_knn_match).  Vocabularies: the reference's own (x_multi_agent_amd/data/vocab_*.npz) and
random trees with uneven child counts / 64-byte descriptors."""
import numpy as np
import pytest

from oracle import ref_pr
from x_multi_agent_amd import engine, place, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(4, 0, 4)
    yield e
    e.close()


VOCS = {
    "visual": lambda: place.load_vocabulary("visual"),
    "thermal": lambda: place.load_vocabulary("thermal"),
    "k10_L3_pruned": lambda: synth.make_vocabulary(10, 3, 32, seed=3, prune=0.35),
    "k2_L6_64B": lambda: synth.make_vocabulary(2, 6, 64, seed=4, prune=0.2),
    "k5_L1": lambda: synth.make_vocabulary(5, 1, 32, seed=5),
}


@pytest.mark.parametrize("vname", list(VOCS))
def test_vlad_bit_exact(eng, vname):
    v = VOCS[vname]()
    voc = ref_pr.Vocabulary(v)
    db = place.Database(eng, v, 0.5, max_desc=2048)
    assert db.vlad_bytes == voc.clusters_n * voc.d_length
    nb = voc.d_length
    for n, seed in ((0, 1), (1, 2), (7, 3), (400, 4), (1300, 5)):
        x = synth.make_descriptors(n, nb, seed=seed)
        if n >= 7:                                   # some exact centroids and duplicates among the queries
            x[0] = voc.get_word(0)
            x[3] = x[5]
        assert np.array_equal(db.compute_vlad(x), ref_pr.compute_vlad(voc, x)), (vname, n)
    # size-independent property at the largest size: OR-linearity
    x = synth.make_descriptors(2048, nb, seed=9)
    full = db.compute_vlad(x)
    assert np.array_equal(full, db.compute_vlad(x[:1000]) | db.compute_vlad(x[1000:]))
    db.close()


def test_capacity_and_argument_errors(eng):
    v = place.load_vocabulary("visual")
    db = place.Database(eng, v, 0.5, max_desc=16)
    with pytest.raises(engine.XkError):
        db.compute_vlad(synth.make_descriptors(17, 32))
    with pytest.raises(ValueError):
        db.find_candidate(0, np.zeros(5, np.uint8))
    with pytest.raises(engine.XkError):
        db.keyframe(0)
    assert db.find_candidate(0, np.zeros(db.vlad_bytes, np.uint8)) == (-1, 0.0, -1)     # empty store
    bad = dict(v)
    bad["desc"] = v["desc"][:, :30]                  # 30-byte descriptors: not a multiple of 4
    with pytest.raises(engine.XkError):
        place.Database(eng, bad, 0.5)
    db.close()


@pytest.mark.parametrize("vname,thr", [("visual", 0.55), ("k10_L3_pruned", 0.9), ("k2_L6_64B", 0.7)])
def test_database_sequence_matches_oracle(eng, vname, thr):
    import torch
    v = VOCS[vname]()
    voc = ref_pr.Vocabulary(v)
    nb = voc.d_length
    pay_n, trk_n = 37, 11
    db = place.Database(eng, v, thr, payload_doubles=pay_n, tracks_doubles=trk_n, max_desc=512)
    ora = ref_pr.Database(voc, thr)
    rng = np.random.default_rng(12)
    scenes = [synth.make_descriptors(60 + 5 * s, nb, seed=100 + s) for s in range(4)]
    payloads = {}
    for i in range(22):                              # beyond the 15-keyframe capacity
        d = synth.observe_descriptors(scenes[i % 4], int(rng.integers(0, 12)), seed=i)
        pay = torch.full((pay_n,), float(i), dtype=torch.float64, device="cuda")
        trk = torch.arange(trk_n, dtype=torch.float64, device="cuda") + 100.0 * i
        torch.cuda.synchronize()                     # (torch filled them on ITS stream; the store copies on the engine's)
        db.add_keyframe(d, pay.data_ptr(), trk.data_ptr(), tag=i)
        ora.add_keyframe(ref_pr.Keyframe(d, tag=i))
        payloads[i] = (pay.cpu().numpy(), trk.cpu().numpy())
        assert len(db) == len(ora.keyframes)
        # requests from three agents interleaved with the insertions
        for uav in (1, 2, 1, 7):
            q = synth.observe_descriptors(scenes[int(rng.integers(0, 4))], int(rng.integers(0, 20)), seed=1000 + 7 * i + uav)
            qv = db.compute_vlad(q)
            assert np.array_equal(qv, ref_pr.compute_vlad(voc, q))
            idx, sc, tag = db.find_candidate(uav, qv)
            okf, oidx, osc = ora.find_candidate(uav, qv)
            assert idx == oidx and sc == osc, (i, uav)          # the score is the same double, not merely close
            assert tag == (okf.tag if okf is not None else -1)
            if idx >= 0:
                kf = db.keyframe(idx, want_descriptors=True)
                assert kf["tag"] == tag and kf["n_desc"] == len(okf.descriptors)
                assert np.array_equal(kf["descriptors"], okf.descriptors)
                got_p = torch.empty(pay_n, dtype=torch.float64, device="cuda")
                got_t = torch.empty(trk_n, dtype=torch.float64, device="cuda")
                db.copy_keyframe(idx, got_p.data_ptr(), got_t.data_ptr())      # what a response sends back
                torch.cuda.synchronize()
                assert np.array_equal(got_p.cpu().numpy(), payloads[tag][0])
                assert np.array_equal(got_t.cpu().numpy(), payloads[tag][1])
    assert [k.tag for k in ora.keyframes] == list(range(7, 22))
    db.close()


@pytest.mark.parametrize("nb", [32, 64])
def test_knn_match_bit_exact(eng, nb):
    v = synth.make_vocabulary(3, 2, nb, seed=1)
    db = place.Database(eng, v, 0.5, max_desc=1500)
    rng = np.random.default_rng(5)
    for nq, nt in ((1, 2), (5, 1), (4, 0), (37, 41), (300, 777), (1500, 1500)):
        train = synth.make_descriptors(nt, nb, seed=nq + nt)
        if nt >= 40:
            train[17] = train[4]                     # ties: the earlier train index must come first
            train[33] = train[4]
        src = train if nt else synth.make_descriptors(8, nb, seed=2)
        query = synth.observe_descriptors(src[rng.integers(0, len(src), nq)], 6, seed=nq)
        if nt >= 40:
            query[0] = train[4]
        idx, dist = db.knn_match(query, train)
        oidx, odist = ref_pr.knn2(query, train)
        assert np.array_equal(idx, oidx), (nq, nt)
        both = oidx >= 0
        assert np.array_equal(dist[both], odist[both])
        if nt >= 40:
            assert tuple(idx[0]) == (4, 17) and tuple(dist[0]) == (0, 0)
    db.close()


def test_request_response_round(eng):
    """One request as the reference runs it (vio.cpp:462-496, 498-570): the requester's VLAD picks a keyframe of the
    responder; the returned descriptors are matched against the requester's and classified."""
    v = place.load_vocabulary("visual")
    voc = ref_pr.Vocabulary(v)
    responder = place.Database(eng, v, 0.6, max_desc=512)
    scene_a, scene_b = synth.make_descriptors(90, 32, seed=21), synth.make_descriptors(90, 32, seed=22)
    responder.add_keyframe(synth.observe_descriptors(scene_b, 4, seed=1), tag=100)
    responder.add_keyframe(synth.observe_descriptors(scene_a, 4, seed=2), tag=101)
    mine = synth.observe_descriptors(scene_a, 4, seed=3)            # the requester looks at scene A
    idx, score, tag = responder.find_candidate(3, responder.compute_vlad(mine))
    assert tag == 101 and score > 0.6
    kf = responder.keyframe(idx, want_descriptors=True)
    nn_idx, nn_dist = responder.knn_match(kf["descriptors"], mine)  # query = received, train = current
    good = place.good_matches(nn_idx, nn_dist, 50.0, 0.8)
    assert good == ref_pr.good_matches(*ref_pr.knn2(kf["descriptors"], mine), 50.0, 0.8)
    assert len(good) >= 80 and all(q == t for q, t in good)         # same landmark order on both sides
    cls = place.classify(good, 30, 30, 30, 30)
    assert {c[0] for c in cls} == {"slam", "opp_opp"}               # q == t: SLAM x SLAM and OPP x OPP pairs only
    assert responder.find_candidate(3, responder.compute_vlad(mine))[2] != 101       # not sent twice
    responder.close()


def test_rejected_keyframe_leaves_a_full_database_untouched(eng):
    """xk_pr_add_keyframe validates its descriptors BEFORE it drops the oldest keyframe of a full database."""
    v = VOCS["visual"]()
    nb = ref_pr.Vocabulary(v).d_length
    db = place.Database(eng, v, 0.5, max_desc=64)
    for i in range(15):
        db.add_keyframe(synth.make_descriptors(20, nb, seed=i), tag=i)
    assert len(db) == 15
    with pytest.raises(engine.XkError):
        db.add_keyframe(synth.make_descriptors(65, nb, seed=99), tag=99)     # more than max_desc
    assert len(db) == 15
    assert [db.keyframe(i)["tag"] for i in range(15)] == list(range(15))      # the oldest one is still there
    db.close()
