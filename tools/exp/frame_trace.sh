#!/bin/bash
export XK_LIB_PATH=${XK_LIB_PATH:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/x_multi_agent_amd/lab/libxk.so}   # lab build: env switches, hooks, probes
# Device timeline of the frame loop (resident covariance): kernel and copy intervals of a few frames under rocprofv3.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from x_multi_agent_amd import synth
sc = synth.make_config(4)
N = sc["n_poses_max"]; off = sc["trk_off"]; K = len(off) - 1
parts = [np.array([N, K, 40, 7, 1, sc["sigma_img"]], float), sc["C_q_G"].ravel(), sc["G_p_C"].ravel(), np.diff(off).astype(float), sc["obs_xy"].ravel(), np.asfortranarray(sc["P"]).ravel(order="F")]
np.concatenate(parts).astype("<f8").tofile("/tmp/fl_in.bin")
PY
rm -rf gpurun_out/ft; LD_LIBRARY_PATH=x_multi_agent_amd:/opt/rocm/lib rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ft -o t -- x_multi_agent_amd/xk_frame_loop_example /tmp/fl_in.bin /tmp/fl_out.bin > gpurun_out/ft.log 2>&1
python - <<'PY'
import csv, glob
k = list(csv.DictReader(open(glob.glob("gpurun_out/ft/**/t_kernel_trace.csv", recursive=True)[0])))
m = list(csv.DictReader(open(glob.glob("gpurun_out/ft/**/t_memory_copy_trace.csv", recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in k]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:20] + " " + r.get("Bytes", r.get("Size", ""))) for r in m]
ev.sort()
# frames start at every xk_msckf_feature; print the frame before last: everything between two feature kernels
idx = [i for i, e in enumerate(ev) if e[2].startswith("xk_msckf_feature")]
a, b = idx[-3], idx[-2]
# back up to the first event after the previous frame's last kernel (the gemm)
start = a
while start > 0 and not ev[start - 1][2].startswith("xk_gemm"): start -= 1
t0 = ev[start][0]
prev_end = ev[start - 1][1] if start > 0 else t0
print(f"idle before this frame's first device op: {(t0 - prev_end)/1e3:.1f} us")
for e in ev[start:b]:
    if e[0] >= ev[b][0]: break
    print(f"{(e[0]-t0)/1e3:9.1f} {(e[1]-e[0])/1e3:8.1f}  {e[2]}")
PY
