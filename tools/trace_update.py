"""Per-launch durations and gaps of one update from a rocprofv3 kernel trace (gpurun_out/ks/k_kernel_trace.csv by
default, written by tools/kstats.sh):  python tools/trace_update.py [trace.csv]"""
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ks/k_kernel_trace.csv"
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
feat = [i for i, r in enumerate(rows) if "msckf_feature" in r["Kernel_Name"]]
i0, i1 = feat[-2], feat[-1]
prev = None
for r in rows[i0:i1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev) / 1e3 if prev else 0.0
    print(f'{r["Kernel_Name"][:44]:44s} {(en - st) / 1e3:7.2f} us   gap {gap:5.2f}')
    prev = en
print(f"update span {(int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3:.1f} us (includes the host-side gap before the next update)")
