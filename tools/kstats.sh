#!/bin/bash
# usage: tools/kstats.sh  -> per-kernel average durations of one bench run (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o k -- python bench.py --steps 20 --warmup 3 --no-cpu --no-frame-loop > gpurun_out/ks.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/ks/k_kernel_stats.csv")))
for r in rows[:7]: print(f"{r['Name'][:52]:52s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
grep "^{" gpurun_out/ks.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['stages_ms'])"
