#!/bin/bash
# QR-stage time of the compression schedules on the same box, back to back (HIP-event stage times of xk_bench_staged)
for rep in 1 2; do
for v in "XK_CAQR_RESIDENT=0" "XK_CAQR_RESIDENT=1" "XK_CAQR_RESIDENT=1 XK_CAQR_BLOCKED=1" "XK_CAQR_RESIDENT=0 XK_CAQR_PERSIST=1"; do
  env $v python - "$v" <<'PY'
import sys
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
sc = synth.make_config(4)
eng = engine.Engine(30, 0, 400)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 5, 50)
qr = sum(v["ms"] for k, v in t["stages"].items() if "caqr" in k)
print(f"{sys.argv[1]:45s} QR {qr:.4f} ms  total {t['total_ms']:.4f} ms")
PY
done; done
