#!/bin/bash
# same box, several builds of libxk.so back to back (XK_LIB_PATH): bash tools/exp/ab_libs.sh lib1.so lib2.so ...
for rep in 1 2; do for lib in "$@"; do
XK_LIB_PATH=$lib python - "$lib" <<'PY'
import sys
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
sc = synth.make_config(4)
eng = engine.Engine(30, 0, 400)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 5, 50)
qr = sum(v["ms"] for k, v in t["stages"].items() if "caqr" in k)
print(f"{sys.argv[1]:44s} QR {qr:.4f} ms  total {t['total_ms']:.4f} ms")
PY
done; done
