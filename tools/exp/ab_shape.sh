#!/bin/bash
# same box, several builds of libxk.so on one synthetic shape: bash tools/exp/ab_shape.sh N K M lib1.so lib2.so ...
N=$1; K=$2; M=$3; shift 3
for rep in 1 2; do for lib in "$@"; do
XK_LIB_PATH=$lib python - "$lib" $N $K $M <<'PY'
import sys
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
N, K, M = (int(v) for v in sys.argv[2:5])
sc = synth.make_scenario(N, K, M, seed=4242)
eng = engine.Engine(N, M, K)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
qr = sum(v["ms"] for k, v in t["stages"].items() if "caqr" in k)
print(f"shape N={N} K={K} M={M} {sys.argv[1]:36s} QR {qr:.4f} ms  total {t['total_ms']:.4f} ms  rows {t['rows_stacked']}", eng.caqr_status()["schedule"])
PY
done; done
