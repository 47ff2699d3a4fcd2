#!/bin/bash
export XK_LIB_PATH=${XK_LIB_PATH:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/x_multi_agent_amd/lab/libxk.so}   # lab build: env switches, hooks, probes
# kernel trace of ONE device-resident CI round at 8 agents (tools/exp/ci_round_trace.py under rocprofv3): start, duration, name
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ci
cd $R
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ci -o p -- python $R/tools/exp/ci_round_trace.py > $R/gpurun_out/ci_round_trace.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ci/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("xk_ci_gather")]
start = idx[-2] if len(idx) >= 2 else idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
PY
