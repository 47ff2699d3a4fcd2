#!/bin/bash
# same box, several builds of libxk.so on one BASELINE config: bash tools/exp/ab_cfg.sh <config> lib1.so lib2.so ...
CFG=$1; shift
for rep in 1 2; do for lib in "$@"; do
XK_LIB_PATH=$lib python - "$lib" "$CFG" <<'PY'
import sys
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
sc = synth.make_config(int(sys.argv[2]))
N = sc["n_poses_max"]; K = len(sc["trk_off"]) - 1; M = len(sc.get("slam_anchor_idxs", []))
eng = engine.Engine(N, M, K)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
qr = sum(v["ms"] for k, v in t["stages"].items() if "caqr" in k)
print(f"cfg{sys.argv[2]} {sys.argv[1]:40s} QR {qr:.4f} ms  total {t['total_ms']:.4f} ms")
PY
done; done
