#!/bin/bash
# wide geometry at fewer rows per lane: N=30, K=165, M=50 (9 505 rows incl. rejected tracks) with several builds of libxk.so
for rep in 1 2; do for lib in "$@"; do
XK_LIB_PATH=$lib python - "$lib" <<'PY'
import sys
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
sc = synth.make_scenario(30, 165, 50, seed=4242)
eng = engine.Engine(30, 50, 165)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
qr = sum(v["ms"] for k, v in t["stages"].items() if "caqr" in k)
print(f"wide165 {sys.argv[1]:40s} QR {qr:.4f} ms  total {t['total_ms']:.4f} ms", eng.caqr_status())
PY
done; done
