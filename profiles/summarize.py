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
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)      # a directory may hold older runs too
here = os.path.dirname(os.path.abspath(__file__))

rows = list(csv.DictReader(open(newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))))
with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "nerf_" in r["Name"]:
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

out = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for r in csv.DictReader(open(newest(os.path.join(d, "*", "*_counter_collection.csv")))):
        if "nerf_" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"], r["Counter_Name"])
        out.setdefault(key, []).append(float(r["Counter_Value"]))
if out:          # (trace-only collections have no counter passes: no empty file)
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

# traffic_<prec>.json: what bench.py reports as roofline.traffic (per launch of the dominant kernel, averaged over the
# coarse and fine launches exactly as `rocprofv3 --stats` averages their durations)
prec = tag.split("_", 1)[1] if tag.startswith("r01_") or tag[:1] == "r" else None
kname = {"f32": "nerf_mlp_f32_kernel", "f16": "nerf_mlp_f16_kernel", "f32x": "nerf_mlp_f32x_kernel"}.get(prec)
if kname:
    import json
    per, clk_num, clk_den = {}, 0.0, 0.0
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for r in csv.DictReader(open(newest(os.path.join(d, "*", "*_counter_collection.csv")))):
            if kname not in r["Kernel_Name"]:
                continue
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                clk_num += float(r["Counter_Value"])
                clk_den += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    mean = lambda k: sum(per[k]) / len(per[k])
    stats = [r for r in rows if kname in r["Name"]]
    out_j = {
        "kernel": kname, "workload": "800x800, 64+128, 1 GPU", "launches_per_frame": 2,
        "fetch_bytes_per_launch": mean("FETCH_SIZE") * 1024 * 2, "write_bytes_per_launch": mean("WRITE_SIZE") * 1024,
        "traffic_bytes_per_launch": mean("FETCH_SIZE") * 1024 * 2 + mean("WRITE_SIZE") * 1024,
        "rocprof_avg_launch_ms": float(stats[0]["AverageNs"]) / 1e6 if stats else None,
        "mfma_busy_frac": sum(per["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(per["GRBM_GUI_ACTIVE"]) * 128.0),
        "clock_ghz": clk_num / clk_den / 8.0,          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        "source": f"profiles/{tag}_pmc.csv, profiles/{tag}_kernel_stats.csv (profiles/collect.sh: rocprofv3 --pmc in separate "
                  "passes; FETCH_SIZE KiB x2 gfx950 wide-read correction, WRITE_SIZE KiB exact; SQ_VALU_MFMA_BUSY_CYCLES / "
                  "(GRBM_GUI_ACTIVE x 128); the PMC passes run slower clocks than the timing pass)",
    }
    json.dump(out_j, open(os.path.join(here, f"traffic_{prec}.json"), "w"), indent=1)
    print("wrote", f"traffic_{prec}.json", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out_j.items() if k != "source"})
print("wrote", tag)
