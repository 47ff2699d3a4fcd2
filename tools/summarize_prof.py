"""Distils rocprofv3 CSV output (gpurun_out/prof_<tag>[_cfgN]_*, written by tools/profile_round.sh) into the tracked
summaries under profiles/.

    python tools/summarize_prof.py r02 [config]
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = sys.argv[2] if len(sys.argv) > 2 else "4"
suf = "" if cfg == "4" else f"_cfg{cfg}"
os.makedirs(PROF, exist_ok=True)


def csrc_sha16():
    """Hash of every kernel source: bench.py refuses PMC numbers taken from other kernels than the ones it runs."""
    hh = hashlib.sha256()
    d = os.path.join(ROOT, "x_multi_agent_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            hh.update(open(os.path.join(d, f), "rb").read())
    return hh.hexdigest()[:16]


def kname(full):
    """Kernel name without return type, template arguments and parameter list."""
    n = full.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0].strip() if n.startswith("xk_") else n


def find(d, pat):
    return glob.glob(os.path.join(OUT, f"prof_{tag}{suf}_{d}", "**", pat), recursive=True)


avg_ns = {}
stats = find("trace", "*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], os.path.join(PROF, f"{tag}{suf}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
    tot = {}
    for r in rows:      # (instantiations of one kernel -- the geometries of xk_caqr_pipe -- are one entry: call-weighted average)
        e = tot.setdefault(kname(r["Name"]), [0.0, 0])
        e[0] += float(r["TotalDurationNs"])
        e[1] += int(r["Calls"])
    for k, (t, c) in tot.items():
        avg_ns[k] = (t / max(c, 1), c)
    with open(os.path.join(PROF, f"{tag}{suf}_kernel_stats.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --config {cfg} --steps 20 --warmup 3 --no-cpu --no-frame-loop ({tag})\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:16]:
            f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                    f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")


def counter_sums(d, names):
    out = {}
    for path in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] not in names:
                continue
            k = kname(r["Kernel_Name"])
            e = out.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return out


pmc = {}
for d, names in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                 ("sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"]),
                 ("l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    for k, v in counter_sums(d, names).items():
        pmc.setdefault(k, {}).update({n: {"sum": s, "dispatches": c, "per_dispatch": s / max(c, 1)} for n, (s, c) in v.items()})
if pmc:
    updates = 13     # bench --steps 5 --warmup 1 plus bench_staged(2 warm + 5)
    per_kernel = {}
    for k, c in pmc.items():
        if not k.startswith(("xk_", "void xk_")):
            continue
        g = lambda n: c.get(n, {}).get("per_dispatch")
        dur = avg_ns.get(k, (None, 0))[0]
        e = {"avg_us": dur / 1e3 if dur else None}
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            e["fetch_bytes_per_launch_corrected"] = 2.0 * g("FETCH_SIZE") * 1024.0     # gfx950: FETCH_SIZE = half the bytes
            e["write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0
            if dur:
                e["l2_fabric_GBps"] = (e["fetch_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]) / dur   # bytes / ns
        if g("SQ_BUSY_CYCLES"):
            # raw ratio of two counters with different aggregation (MFMA-busy is summed over SIMDs); normalised below by the
            # same ratio of xk_probe_mfma, a kernel that does nothing but back-to-back v_mfma_f64 (100 % by construction)
            e["mfma_busy_over_sq_busy"] = (g("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0) / g("SQ_BUSY_CYCLES")
        if g("SQ_WAVE_CYCLES"):
            e["wait_any_pct_of_wave_cycles"] = 100.0 * (g("SQ_WAIT_ANY") or 0.0) / g("SQ_WAVE_CYCLES")
            # vector-pipe activity per wave cycle, normalised below by xk_probe_fma (back-to-back v_fma_f64 = 100 %)
            e["valu_active_over_wave_cycles"] = (g("SQ_ACTIVE_INST_VALU") or 0.0) / g("SQ_WAVE_CYCLES")
        if g("TCC_HIT_sum") is not None and (g("TCC_HIT_sum") + (g("TCC_MISS_sum") or 0)) > 0:
            e["l2_hit_pct"] = 100.0 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        per_kernel[k] = e
    sat = per_kernel.get("xk_probe_mfma", {}).get("mfma_busy_over_sq_busy")
    satv = per_kernel.get("xk_probe_fma", {}).get("valu_active_over_wave_cycles")
    for k, e in per_kernel.items():
        if sat and "mfma_busy_over_sq_busy" in e:
            e["mfma_util_pct"] = 100.0 * e["mfma_busy_over_sq_busy"] / sat
        if satv and "valu_active_over_wave_cycles" in e:
            e["valu_util_pct_of_resident_waves"] = 100.0 * e["valu_active_over_wave_cycles"] / satv
    res = {"note": "FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts 64 B per 128-B request "
                   "(MI355X_MICROARCH.md HBM section) -> fetch bytes corrected = 2x.  Both count requests from the L2 to the fabric: "
                   "Infinity-Cache (MALL) hits are included, no counter on this stack separates them from HBM accesses.  "
                   "valu_util_pct_of_resident_waves = SQ_ACTIVE_INST_VALU per SQ_WAVE_CYCLES relative to xk_probe_fma (waves that do nothing but "
                   "dependent v_fma_f64): how busy a RESIDENT wave keeps the vector pipe, not a chip-level utilisation.",
           "csrc_sha16": csrc_sha16(), "config": int(cfg), "updates_in_run": updates, "per_kernel": per_kernel, "kernels": pmc}
    qr = [k for k in pmc if "caqr" in k]
    f = sum(pmc[k].get("FETCH_SIZE", {}).get("sum", 0.0) for k in qr) * 1024.0
    w = sum(pmc[k].get("WRITE_SIZE", {}).get("sum", 0.0) for k in qr) * 1024.0
    res["qr_fetch_bytes_per_update_raw"] = f / updates
    res["qr_write_bytes_per_update"] = w / updates
    res["qr_bytes_per_update"] = (2.0 * f + w) / updates
    json.dump(res, open(os.path.join(PROF, f"{tag}{suf}_pmc_traffic.json"), "w"), indent=1)
for name in ("fp64_peak.json",):
    p = os.path.join(OUT, name)
    if os.path.exists(p) and cfg == "4":
        shutil.copy(p, os.path.join(PROF, f"{tag}_{name}"))
p = os.path.join(OUT, f"bench_line_{tag}{suf}.json")
if os.path.exists(p) and os.path.getsize(p):
    shutil.copy(p, os.path.join(PROF, f"{tag}{suf}_bench_under_rocprof.json"))
print(open(os.path.join(PROF, f"{tag}{suf}_kernel_stats.md")).read() if stats else "no stats")
if pmc:
    for k, e in per_kernel.items():
        print(k[:44], {a: (round(b, 2) if isinstance(b, float) else b) for a, b in e.items() if a not in ("fetch_bytes_per_launch_corrected", "write_bytes_per_launch")})
