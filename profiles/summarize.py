#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries kept under profiles/.

    python profiles/summarize.py gpurun_out/prof r01_f32

writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, our kernels only) and
profiles/<tag>_pmc.csv (per-dispatch counters of the separate --pmc passes, with the gfx950
FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied in the *_bytes columns)."""
import csv
import glob
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))

rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0])))
with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "nerf_" in r["Name"]:
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

out = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0])):
        if "nerf_" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"], r["Counter_Name"])
        out.setdefault(key, []).append(float(r["Counter_Value"]))
with open(os.path.join(here, f"{tag}_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "counter", "dispatches", "mean_value", "note"])
    for (k, g, c), v in sorted(out.items()):
        m = sum(v) / len(v)
        note = ""
        if c == "FETCH_SIZE":
            note = f"KiB units; x2 gfx950 wide-read correction -> {m * 1024 * 2 / 1e6:.1f} MB/dispatch"
        elif c == "WRITE_SIZE":
            note = f"KiB units -> {m * 1024 / 1e6:.1f} MB/dispatch"
        w.writerow([k, g, c, len(v), f"{m:.0f}", note])
print("wrote", tag)
