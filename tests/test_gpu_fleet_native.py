"""The native exchange of the CI step (include/xk_fleet.h on RCCL) with the one GPU a test box has: a communicator of
one rank on the engine's stream -- ncclAllGather and a grouped send/recv to itself move the packed SimpleState payload
between device buffers bit for bit -- and bench.py's N > 1 path with several ranks sharing the GPU (gloo exchange)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(__file__), "..")


def test_fleet_exchange_on_one_rank(xk):
    import torch
    F = C.CDLL(os.path.join(ROOT, "x_multi_agent_amd", "libxk_fleet.so"))
    F.xk_fleet_last_error.restype = C.c_char_p
    sc = synth.make_config(1)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.Engine(N, 0, K)
    eng.stage(sc)
    uid = (C.c_ubyte * 128)()
    assert F.xk_fleet_unique_id(uid) == 0
    fl = C.c_void_p()
    assert F.xk_fleet_create(eng.h, uid, 1, 0, C.byref(fl)) == 0
    assert F.xk_fleet_world(fl) == 1 and F.xk_fleet_rank(fl) == 0
    n = eng.payload_doubles()
    send = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    recv = torch.full((n,), -1.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    dyn = np.zeros(16); dyn[9] = 1.0
    eng.pack_payload_into(7, 3.5, dyn, send.data_ptr())            # queued on the engine's stream; the gather follows it there
    dp = lambda t: C.cast(C.c_void_p(t.data_ptr()), C.POINTER(C.c_double))
    assert F.xk_fleet_all_gather(fl, dp(send), dp(recv), C.c_long(n)) == 0, F.xk_fleet_last_error(fl)
    assert F.xk_fleet_wait(fl) == 0
    assert torch.equal(send, recv) and float(recv[0]) == 7.0 and float(recv[1]) == 3.5
    back = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    assert F.xk_fleet_send_recv(fl, dp(recv), C.c_long(n), 0, dp(back), C.c_long(n), 0) == 0, F.xk_fleet_last_error(fl)
    assert F.xk_fleet_wait(fl) == 0
    assert torch.equal(back, send)
    assert F.xk_fleet_send_recv(fl, dp(recv), C.c_long(n), 1, dp(back), C.c_long(n), 0) == 1      # peer outside the world
    assert F.xk_fleet_destroy(fl) == 0
    eng.close()


@pytest.mark.parametrize("config,ranks", [(4, 2), (5, 2)])
def test_bench_multi_rank_path_on_one_gpu(config, ranks):
    """bench.py --gpus N as the driver launches it (torch.distributed.run, one process per rank), with the ranks sharing
    the one GPU and the exchange over gloo: the N > 1 plumbing -- payload packing, all-gather / request-response, the
    device CI round -- runs end to end and fuses tracks."""
    env = dict(os.environ, XK_BENCH_BACKEND="gloo", XK_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(29650 + config), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "24", "--warmup", "2",
           "--config", str(config), "--no-cpu", "--no-frame-loop"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["ci_rounds_rank0"] >= 2 and d["config"]["ci_fused_rank0"] > 0
