"""Times whole filter frames through the C++ mirror (host/examples/frame_loop_main.cpp): 7 IMU steps, manage(), the visual
update, State::correct and the re-propagation, with the covariance resident on the device / owned by the State.

    python tools/bench_frame_loop.py [config] [frames]
"""
import os, subprocess, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from x_multi_agent_amd import synth

PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "x_multi_agent_amd")


def run(sc, frames, imu_per_frame, mode):
    N = sc["n_poses_max"]; off = sc["trk_off"]; K = len(off) - 1
    parts = [np.array([N, K, frames, imu_per_frame, mode, sc["sigma_img"]], float), sc["C_q_G"].ravel(), sc["G_p_C"].ravel(),
             np.diff(off).astype(float), sc["obs_xy"].ravel(), np.asfortranarray(sc["P"]).ravel(order="F")]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        np.concatenate(parts).astype("<f8").tofile(fin)
        env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([os.path.join(PKG, "xk_frame_loop_example"), fin, fout], capture_output=True, text=True, env=env, timeout=900)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        out = np.fromfile(fout, dtype="<f8")
    n = 15 + 6 * N
    return out[n * n + 7 * N + 16:], r.stdout.strip()


if __name__ == "__main__":
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    sc = synth.make_config(cfg)
    for mode, name in ((1, "resident, IMU steps composed"), (2, "resident, one device step per IMU step"), (0, "State owns the covariance")):
        ms, log = run(sc, frames, 7, mode)
        ms = ms[20:]
        print(f"{name:42s} median {np.median(ms):.4f} ms  mean {ms.mean():.4f}  min {ms.min():.4f}  p95 {np.percentile(ms, 95):.4f}   ({log})")
