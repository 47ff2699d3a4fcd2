#!/bin/bash
# same box, several builds of libxk.so back to back at BASELINE config 2: bash tools/exp/ab_cfg2.sh lib1.so lib2.so ...
for rep in 1 2; do for lib in "$@"; do
XK_LIB_PATH=$lib python - "$lib" <<'PY'
import sys
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
sc = synth.make_config(2)
eng = engine.Engine(30, 50, 200)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 5, 50)
qr = sum(v["ms"] for k, v in t["stages"].items() if "caqr" in k)
print(f"{sys.argv[1]:44s} QR {qr:.4f} ms  total {t['total_ms']:.4f} ms leaves {t['n_leaf']}")
PY
done; done
