#!/bin/bash
# QR stage alone (build once, then xk_qr_compress repeated; host-timed, one synchronisation per call) for several builds of
# libxk.so -- usable with timing probes whose results are wrong (the Kalman stage would refuse them): bash tools/exp/ab_qr_only.sh <config> libs...
CFG=$1; shift
for rep in 1 2; do for lib in "$@"; do
XK_LIB_PATH=$lib python - "$lib" "$CFG" <<'PY'
import sys, time
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[2])
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
eng = engine.Engine(N, M, K)
eng.stage(sc)
ts = []
for i in range(60):
    eng.msckf_build(sc["sigma_img"])
    t0 = time.perf_counter(); eng.qr_compress(False); ts.append(time.perf_counter() - t0)
ts = sorted(ts[10:])
print(f"cfg{cfg} {sys.argv[1]:36s} xk_qr_compress (launch + sync) median {ts[len(ts)//2]*1e3:.4f} ms  min {ts[0]*1e3:.4f} ms", eng.caqr_status()["schedule"])
PY
done; done
