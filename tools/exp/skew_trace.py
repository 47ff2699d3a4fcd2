"""Skew of the hand-offs of the single launch across workgroups (XK_CAQR_PERSIST_DBG=1, lab build): per panel, when each
first-level workgroup had its pending strip / had its phase-0 root rows out, and when each last-level workgroup had the roots."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("XK_CAQR_PERSIST_DBG", "1")
os.environ.setdefault("XK_PIPE_SPLIT", "3")
import numpy as np
from x_multi_agent_amd import engine, synth
sc = synth.make_config(4)
N, K, M = synth.CONFIGS[4]
eng = engine.LabEngine(N, M, K)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 10)
NW = 65536
out = (C.c_longlong * NW)()
eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
w = np.array(list(out), dtype=np.int64)
T = w[:512].reshape(32, 16)
us = lambda x: x / 100.0
M1 = w[4096:4096 + 96 * 32].reshape(96, 32)       # [wg][0..15 pending in | 16..31 phase-0 rows out]
Lw = w[8192:8192 + 7 * 64].reshape(7, 64)          # [lidx][0..15 ph0 in | 16..31 ph1 in | 32..47 out]
for k in range(1, 11):
    t0 = T[k, 0]
    pin = M1[:, k]; pout = M1[:, 16 + k]
    pin = pin[pin > 0]; pout = pout[pout > 0]
    l0 = Lw[:, k]; l1 = Lw[:, 16 + k]; lo = Lw[:, 32 + k]; lop = Lw[:, 32 + k - 1]
    print(f"panel {k:2d}: L(k-1) out {us(lop.min()-t0):6.2f}..{us(lop.max()-t0):6.2f} | M1 pending in {us(pin.min()-t0):6.2f}..{us(pin.max()-t0):6.2f} | "
          f"M1 ph0 out {us(pout.min()-t0):6.2f}..{us(pout.max()-t0):6.2f} | L ph0 in {us(l0.min()-t0):6.2f}..{us(l0.max()-t0):6.2f} | "
          f"L ph1 in {us(l1.min()-t0):6.2f}..{us(l1.max()-t0):6.2f} | L out {us(lo.min()-t0):6.2f}..{us(lo.max()-t0):6.2f}")
MP = w[16640:16640 + 96 * 16 * 4].reshape(96, 16, 4)          # [first-level wg][panel][phase] root rows of the phase out
LP = w[32768:32768 + 7 * 16 * 4 * 2].reshape(7, 16, 4, 2)      # [last-level wg][panel][phase][in, done]
print("per phase: first level out (earliest .. latest workgroup) -> last level in (earliest .. latest) / done, us after XCD 0's tile started the panel")
for k in (3, 6, 9):
    t0 = T[k, 0]
    cells = []
    for q in range(4):
        m = MP[:, k, q]; m = m[m > 0]
        li = LP[:, k, q, 0]; li = li[li > 0]; ld = LP[:, k, q, 1]; ld = ld[ld > 0]
        cells.append(f"q{q}: M1 {us(m.min()-t0):5.1f}..{us(m.max()-t0):5.1f} -> L in {us(li.min()-t0):5.1f}..{us(li.max()-t0):5.1f} done {us(ld.max()-t0):5.1f}")
    print(f"panel {k}: " + " | ".join(cells))
TS = w[12288:12288 + 8 * 32].reshape(8, 32)
print("per XCD, us after XCD 0's tile started the panel: tile start | group 0 / 1: pending in, phase-0 rows out (slowest item)")
for k in (3, 6, 9):
    t0 = TS[0, k]
    row = []
    for x in range(8):
        cells = []
        for g in range(2):
            wg = M1[(x * 2 + g) * 6:(x * 2 + g) * 6 + 6]
            pin = wg[:, k][wg[:, k] > 0]; pout = wg[:, 16 + k][wg[:, 16 + k] > 0]
            cells.append(f"{us(pin.max()-t0) if len(pin) else float('nan'):5.1f}/{us(pout.max()-t0) if len(pout) else float('nan'):5.1f}")
        row.append(f"X{x} {us(TS[x,k]-t0):5.1f} | " + " ".join(cells))
    print(f"panel {k}: " + "  ||  ".join(row))
eng.close()
