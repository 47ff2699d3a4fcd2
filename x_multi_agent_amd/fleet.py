"""One-agent-per-GPU fleet plumbing: payload layout and the CI exchange step.

Each agent is a complete filter (reference: `VIO` owns its `Ekf`, state and P;
include/x/vio/vio.h:225-247); the only coupling is the CI message, a
`SimpleState` snapshot (include/x/ekf/simple_state.h:33-35, assembled at
src/x/vio/vio.cpp:447-450).  Agents therefore shard one per rank with NO
data-path collective inside an update; every `ci_every` updates the payloads
are exchanged:
  broadcast mode   (VIO::getDataToSend, vio.cpp:440-451)     -> one all-gather
  request/response (VIO::processOtherRequests, vio.cpp:462-496) -> send/recv pairs: the requester's binary
                   VLAD goes out, the responder's best keyframe (place.Database, xk_pr_*) comes back
The same functions run over RCCL (backend "nccl", GPU tensors) in bench.py and
over gloo (CPU tensors) in tests/test_fleet_gloo.py.

Payload (all doubles, matches xk_payload_doubles / xk_pack_payload in xk.h):
  hdr[8]  {agent_id, timestamp, N, M, n, n_poses_valid, 0, 0}
  dyn[16] p,v,q(xyzw),b_w,b_a        (State::getDynamicStates, state.cpp:87-99)
  pos[3N] att[4N xyzw] feat[3M] anchors[M]  cov[n*n] column-major
"""
import numpy as np

K_CORE = 15


def payload_layout(N, M):
    n = K_CORE + 6 * N + 3 * M
    o = {}
    o["hdr"] = (0, 8)
    o["dyn"] = (8, 24)
    o["pos"] = (24, 24 + 3 * N)
    o["att"] = (o["pos"][1], o["pos"][1] + 4 * N)
    o["feat"] = (o["att"][1], o["att"][1] + 3 * M)
    o["anchors"] = (o["feat"][1], o["feat"][1] + M)
    o["cov"] = (o["anchors"][1], o["anchors"][1] + n * n)
    o["total"] = o["cov"][1]
    o["n"] = n
    return o


def pack_payload_host(agent_id, timestamp, dyn16, C_q_G, G_p_C, feat, anchors, P, N, M):
    """Host-side packer with the layout of the device packer (xk_pack_payload)."""
    lay = payload_layout(N, M)
    buf = np.zeros(lay["total"])
    npz = len(G_p_C)
    buf[0:8] = [agent_id, timestamp, N, M, lay["n"], npz, 0, 0]
    buf[8:24] = dyn16
    buf[lay["pos"][0]:lay["pos"][0] + 3 * npz] = np.asarray(G_p_C, float).ravel()
    buf[lay["att"][0]:lay["att"][0] + 4 * npz] = np.asarray(C_q_G, float).ravel()
    mcur = 0 if feat is None else len(feat) // 3
    if mcur:
        buf[lay["feat"][0]:lay["feat"][0] + 3 * mcur] = feat
    buf[lay["anchors"][0]:lay["anchors"][1]] = -1.0
    if mcur:
        buf[lay["anchors"][0]:lay["anchors"][0] + mcur] = anchors
    buf[lay["cov"][0]:lay["cov"][1]] = np.asarray(P, float).ravel(order="F")
    return buf


def unpack_payload(buf, N, M):
    """Inverse of the packers -> dict(agent_id, timestamp, n_poses, dyn, C_q_G, G_p_C, feat, anchors, P)."""
    buf = np.asarray(buf, dtype=np.float64)
    lay = payload_layout(N, M)
    assert buf.size == lay["total"] and int(buf[2]) == N and int(buf[3]) == M and int(buf[4]) == lay["n"]
    npz = int(buf[5])
    n = lay["n"]
    anchors = buf[lay["anchors"][0]:lay["anchors"][1]].astype(np.int32)
    return dict(agent_id=int(buf[0]), timestamp=float(buf[1]), n_poses=npz, dyn=buf[8:24].copy(),
                G_p_C=buf[lay["pos"][0]:lay["pos"][0] + 3 * npz].reshape(npz, 3).copy(),
                C_q_G=buf[lay["att"][0]:lay["att"][0] + 4 * npz].reshape(npz, 4).copy(),
                feat=buf[lay["feat"][0]:lay["feat"][1]].copy(), anchors=anchors,
                P=buf[lay["cov"][0]:lay["cov"][1]].reshape(n, n, order="F").copy())


class Exchange:
    """CI message exchange over torch.distributed (RCCL on GPUs, gloo on CPU).

    `world` is the size of the fleet the buffers are laid out for, `real_world` the number of ranks the process group really
    has.  They differ in a DRY RUN (bench.py --dry-run-ranks N: one process walks rank `rank` of a fleet of N whose other
    agents' messages are already in the receive buffers): every collective then runs on the real one-rank communicator --
    this rank's own slot still travels through RCCL (all_gather_into_tensor with one rank) -- and `peer_answer` stands in for
    the point-to-point partner."""

    def __init__(self, dist, world, rank, payload_doubles, device, dtype=None, real_world=None):
        import torch
        self.dist, self.world, self.rank, self.n = dist, world, rank, payload_doubles
        self.real_world = world if real_world is None else real_world
        self.dry = self.real_world != world
        self.torch = torch
        dtype = dtype or torch.float64           # torch.uint8 for the binary VLAD of a request
        self.send = torch.zeros(payload_doubles, dtype=dtype, device=device)
        self.recv = torch.zeros(payload_doubles * max(world, 1), dtype=dtype, device=device)
        self.peer_answer = None                  # dry run: callable(responder) -> tensor the virtual responder sends back

    def wire(self, t):
        """Dry run: pass a message through the real (one-rank) communicator and hand back what arrived."""
        out = self.torch.empty_like(t)
        if self.dist is not None:
            self.dist.all_gather_into_tensor(out, t.contiguous())
        else:
            out.copy_(t)
        return out

    def all_gather(self):
        """Broadcast mode: every agent receives every agent's payload (config 4)."""
        if self.dry:
            mine = self.recv.view(self.world, self.n)[self.rank]     # the other slots hold the virtual agents' payloads
            if self.dist is not None:
                self.dist.all_gather_into_tensor(mine, self.send)
            else:
                mine.copy_(self.send)
        elif self.world == 1:
            self.recv.copy_(self.send)
        else:
            self.dist.all_gather_into_tensor(self.recv, self.send)
        return self.recv.view(max(self.world, 1), self.n)

    def request_response(self, requests):
        """Request/response mode (config 5): `requests` = list of (requester, responder) pairs for
        this tick; the responder's payload goes point-to-point to the requester.  Returns
        {responder: tensor} for the pairs in which this rank is the requester."""
        got = {}
        ops = []
        for req, rsp in requests:
            if req == rsp:
                continue
            if self.dry:
                if self.rank == req and self.peer_answer is not None:
                    got[rsp] = self.wire(self.peer_answer(rsp))
                continue
            if self.rank == rsp:
                ops.append(self.dist.P2POp(self.dist.isend, self.send, req))
            if self.rank == req:
                buf = self.torch.empty_like(self.send)
                got[rsp] = buf
                ops.append(self.dist.P2POp(self.dist.irecv, buf, rsp))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        return got


def request_round(vex, rex, requests, my_vlad, answer):
    """One request/response tick of the reference's REQUEST_COMM mode.  Every requester's binary VLAD
    (VIO::getDescriptors, vio.cpp:455-460 -> Database::computeVLAD) travels to its responder over `vex`; on the
    responder `answer(requester, vlad_tensor)` runs the keyframe search (VIO::processOtherRequests, vio.cpp:462-496)
    and fills `rex.send` -- word 0 = 1.0 if a keyframe goes back, followed by its SimpleState payload and tracks;
    the responses travel back over `rex` (-> VIO::processOtherMeasurements, vio.cpp:498-570).
    requests: [(requester, responder)], at most one request per responder and tick.
    Returns {responder: response tensor} for the requests this rank made."""
    vex.send.copy_(my_vlad)
    asked = vex.request_response([(rsp, req) for req, rsp in requests])   # requests flow requester -> responder
    if len(asked) > 1:
        raise ValueError("more than one request per responder and tick")
    rex.send[0] = 0.0
    for requester, vlad in asked.items():
        answer(requester, vlad)
    return rex.request_response(requests)


def ring_requests(world, tick):
    """Deterministic 5 Hz request pattern: at tick t agent a asks agent (a + 1 + t mod (world-1)) mod world."""
    if world < 2:
        return []
    return [(a, (a + 1 + tick % (world - 1)) % world) for a in range(world)]


def shared_scenario(synth, config, rank, **kw):
    """Agent `rank`'s synthetic problem for the multi-agent configs: every agent observes agent 0's
    landmark set (ground-truth association, mirroring the reference's GT_DEBUG build) from its own
    arc of the circle, with its own noise, window error and prior."""
    N, K, M = synth.CONFIGS[config]
    if rank == 0:
        return synth.make_config(config, agent_id=0, **kw)
    lm = synth.make_config(config, agent_id=0, **kw)["landmarks_true"]
    return synth.make_scenario(N, K, M, seed=synth.seed_for(config, rank), agent_offset=0.02 * rank, landmarks=lm, **kw)


def pack_tracks(sc, n_tracks, N):
    """Observations of the first n_tracks tracks, padded to N poses each: [n_tracks, 1 + 2N] (length first)."""
    out = np.zeros((n_tracks, 1 + 2 * N))
    off = sc["trk_off"]
    for j in range(n_tracks):
        o = sc["obs_xy"][off[j]:off[j + 1]]
        out[j, 0] = len(o)
        out[j, 1:1 + 2 * len(o)] = o.ravel()
    return out


def unpack_tracks(buf, N):
    buf = np.asarray(buf).reshape(-1, 1 + 2 * N)
    return [row[1:1 + 2 * int(row[0])].reshape(int(row[0]), 2).copy() for row in buf]


def ci_round(eng, sc, others, n_tracks, ci_msckf_w):
    """One MSCKF-MSCKF CI round of an agent against the snapshots it received
    (MsckfUpdate::preProcessOneTrack CI block + Updater::applyCI per list entry,
    msckf_update.cpp:96-279, updater.cpp:90-93).  `others`: list of
    dict(C_q_G, G_p_C, P, tracks=[obs per shared track]).  Returns (n_fused, last posterior or None):
    every P_j is built from the same prior and applyCI overwrites P each time."""
    N = sc["n_poses_max"]
    off = sc["trk_off"]
    fused, last = 0, None
    for j in range(n_tracks):
        trk = sc["obs_xy"][off[j]:off[j + 1]]
        matches = [dict(obs=o["tracks"][j], q_list=o["C_q_G"], p_list=o["G_p_C"], P=o["P"], n_poses_max=N)
                   for o in others]
        r = eng.msckf_ci_track(trk, sc["C_q_G"], sc["G_p_C"], sc["P"], N, sc["sigma_img"], matches, ci_msckf_w)
        if r["ci"] is not None:
            c = r["ci"]
            last, _corr = eng.apply_ci(c["P_j"], c["H"], c["res"], c["S"])
            fused += 1
    return fused, last


def ci_round_device(eng, sc, rank, world, payloads, tracks, n_tracks, ci_msckf_w, want_corrections=False):
    """The same CI round with everything the other agents sent left in device memory.
    payloads: torch CUDA tensor [world, payload_doubles] (the all-gather output); tracks: torch CUDA tensor
    [world, n_tracks * (1 + 2N)] (pack_tracks of every agent).  The engine must have agent `rank`'s problem staged
    (its shared tracks are its first n_tracks staged tracks).  Only the few header / length words come to the host."""
    N = sc["n_poses_max"]
    # the few words the host needs -- every agent's track lengths and window size -- in ONE device-to-host copy
    words = payloads.new_empty(world * (n_tracks + 1))
    words[:world * n_tracks].view(world, n_tracks).copy_(tracks.view(world, n_tracks, 1 + 2 * N)[:, :, 0])
    words[world * n_tracks:].copy_(payloads[:, 5])
    words = words.to("cpu").numpy()
    tl = np.ascontiguousarray(words[:world * n_tracks].reshape(world, n_tracks)).astype(np.int32)
    nv = words[world * n_tracks:].astype(np.int32)
    nv[rank] = len(sc["G_p_C"])
    return eng.ci_round_device(payloads.data_ptr(), payloads.shape[1], world, rank, tracks.data_ptr(), n_tracks, tl, nv,
                               np.arange(n_tracks, dtype=np.int32), sc["sigma_img"], ci_msckf_w, want_corrections)
