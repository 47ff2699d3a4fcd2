"""Distils rocprofv3 CSV output (gpurun_out/prof_r01_*) into the tracked summaries under profiles/.

    python tools/summarize_prof.py [tag]
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs(PROF, exist_ok=True)

stats = glob.glob(os.path.join(OUT, "prof_r01_trace", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(PROF, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(PROF, f"{tag}_kernel_stats.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu ({tag})\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:14]:
            f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                    f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")


def counter_sums(d, names):
    """Sum counter values per kernel name over dispatches; returns {kernel: {counter: (sum, dispatches)}}."""
    out = {}
    for path in glob.glob(os.path.join(OUT, d, "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] not in names:
                continue
            k = r["Kernel_Name"].split("(")[0]
            e = out.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return out


pmc = {}
for d, names in (("prof_r01_fetch", ["FETCH_SIZE"]), ("prof_r01_write", ["WRITE_SIZE"]),
                 ("prof_r01_sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU",
                                  "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"])):
    for k, v in counter_sums(d, names).items():
        pmc.setdefault(k, {}).update({n: {"sum": s, "dispatches": c, "per_dispatch": s / max(c, 1)} for n, (s, c) in v.items()})
if pmc:
    # bench --steps 5 --warmup 1 plus bench_staged(2 warm + 5) = 13 updates in the PMC runs
    updates = 13
    res = {"note": "FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts 64 B per 128-B "
                   "request for wide coalesced reads (MI355X_MICROARCH.md HBM section) -> fetch_bytes_corrected = 2x.",
           "updates_in_run": updates, "kernels": pmc}
    tsqr = [k for k in pmc if "caqr" in k]
    f = sum(pmc[k].get("FETCH_SIZE", {}).get("sum", 0.0) for k in tsqr) * 1024.0
    w = sum(pmc[k].get("WRITE_SIZE", {}).get("sum", 0.0) for k in tsqr) * 1024.0
    res["qr_fetch_bytes_per_update_raw"] = f / updates
    res["qr_write_bytes_per_update"] = w / updates
    res["qr_bytes_per_update"] = (2.0 * f + w) / updates
    json.dump(res, open(os.path.join(PROF, f"{tag}_pmc_traffic.json"), "w"), indent=1)
for name in ("fp64_peak.json",):
    p = os.path.join(OUT, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(PROF, f"{tag}_{name}"))
print(open(os.path.join(PROF, f"{tag}_kernel_stats.md")).read() if stats else "no stats")
