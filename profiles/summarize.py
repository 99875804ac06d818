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


def kname_of(full):
    """nerf_xxx_kernel<template args> out of a demangled name (anonymous-namespace prefixes and argument lists dropped)."""
    import re
    m = re.search(r"nerf_\w+(<[^(]*>)?", full)
    return m.group(0) if m else full[:48]


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
        key = (kname_of(r["Kernel_Name"]), r["Grid_Size"], r["Counter_Name"])
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

# ---- per-kernel attribution of the SQ counters (round-2 VERDICT item 5): where the non-MFMA cycles go
sq = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_sq"))):
    for r in csv.DictReader(open(newest(os.path.join(d, "*", "*_counter_collection.csv")))):
        if "nerf_" not in r["Kernel_Name"]:
            continue
        k = kname_of(r["Kernel_Name"])
        e = sq.setdefault(k, {"dispatches": set()})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["dispatches"].add(r["Dispatch_Id"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            e["_ns"] = e.get("_ns", 0.0) + float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
if sq:
    with open(os.path.join(here, f"{tag}_sq_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "clock_ghz", "mfma_busy_frac_of_simd_cycles", "wave_wait_any_frac", "wave_wait_inst_any_frac",
                    "valu_inst_per_wave_cycle", "lds_bank_conflict_cycles_per_busy_cycle",
                    "note: fractions of SQ_WAVE_CYCLES except mfma_busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128 SIMD-slots))"])
        for k, e in sorted(sq.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
            g = e.get("GRBM_GUI_ACTIVE", 0.0)
            wc = e.get("SQ_WAVE_CYCLES", 0.0)
            if g <= 0 or wc <= 0:
                continue
            w.writerow([k, len(e["dispatches"]), f"{g / e['_ns'] / 8.0:.3f}" if e.get("_ns") else "",
                        f"{e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (g * 128.0):.4f}",
                        f"{e.get('SQ_WAIT_ANY', 0.0) / wc:.4f}", f"{e.get('SQ_WAIT_INST_ANY', 0.0) / wc:.4f}",
                        f"{e.get('SQ_ACTIVE_INST_VALU', 0.0) / wc:.4f}",
                        f"{e.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(1.0, e.get('SQ_BUSY_CYCLES', 0.0)):.5f}", ""])
    print("wrote", f"{tag}_sq_summary.csv")

# ---- training step: HBM bytes per step (all kernels of a step), for bench.py's training.roofline.traffic
if "train" in tag:
    import json
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    per_kernel = {}
    steps_seen = 0
    for d in sorted(glob.glob(os.path.join(src, "pmc_fetch")) + glob.glob(os.path.join(src, "pmc_write"))):
        for r in csv.DictReader(open(newest(os.path.join(d, "*", "*_counter_collection.csv")))):
            if "nerf_" not in r["Kernel_Name"] or r["Counter_Name"] not in tot:
                continue
            v = float(r["Counter_Value"]) * 1024 * (2 if r["Counter_Name"] == "FETCH_SIZE" else 1)
            tot[r["Counter_Name"]] += v
            pk = per_kernel.setdefault(kname_of(r["Kernel_Name"]), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
            pk[r["Counter_Name"]] += v
            if "nerf_adam_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                steps_seen += 1                                   # one Adam launch per step
    steps_seen = max(1, steps_seen)
    prec_t = tag.split("train_", 1)[1]
    out_t = {"workload": "BASELINE configs[2]: 4096 rays/iter, fwd(save)+bwd+Adam, 1 GPU", "steps_profiled": steps_seen,
             "fetch_bytes_per_step": tot["FETCH_SIZE"] / steps_seen, "write_bytes_per_step": tot["WRITE_SIZE"] / steps_seen,
             "traffic_bytes_per_step": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps_seen,
             "per_kernel_bytes_per_step": {k: {"fetch": v["FETCH_SIZE"] / steps_seen, "write": v["WRITE_SIZE"] / steps_seen}
                                           for k, v in sorted(per_kernel.items())},
             "source": f"profiles/{tag}_pmc.csv (profiles/collect.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of "
                       "`bench.py --mode train`; FETCH_SIZE KiB x2 gfx950 wide-read correction, WRITE_SIZE KiB exact; summed over "
                       "every kernel of a step)"}
    json.dump(out_t, open(os.path.join(here, f"traffic_train_{prec_t}.json"), "w"), indent=1)
    print("wrote", f"traffic_train_{prec_t}.json", round(out_t["traffic_bytes_per_step"] / 1e9, 2), "GB/step")

# traffic_<prec>.json: what bench.py reports as roofline.traffic (per launch of the dominant kernel, averaged over the
# coarse and fine launches exactly as `rocprofv3 --stats` averages their durations)
prec = tag.split("_", 1)[1] if (tag[:1] == "r" and "train" not in tag) else None
kname = {"f32": "nerf_mlp_f32_kernel", "f16": "nerf_mlp_f16s_kernel", "f32x": "nerf_mlp_f32x_kernel", "f16m32": "nerf_mlp_f16_kernel"}.get(prec)
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
        # average over every launch of the kernel (both template instances: the density-only coarse launch and the fine launch)
        "rocprof_avg_launch_ms": (sum(float(r["TotalDurationNs"]) for r in stats) / sum(int(r["Calls"]) for r in stats) / 1e6) if stats else None,
        "rocprof_avg_ms_by_instance": {kname_of(r["Name"]): float(r["AverageNs"]) / 1e6 for r in stats},
        "mfma_busy_frac": sum(per["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(per["GRBM_GUI_ACTIVE"]) * 128.0),
        "clock_ghz": clk_num / clk_den / 8.0,          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        "source": f"profiles/{tag}_pmc.csv, profiles/{tag}_kernel_stats.csv (profiles/collect.sh: rocprofv3 --pmc in separate "
                  "passes; FETCH_SIZE KiB x2 gfx950 wide-read correction, WRITE_SIZE KiB exact; SQ_VALU_MFMA_BUSY_CYCLES / "
                  "(GRBM_GUI_ACTIVE x 128); the PMC passes run slower clocks than the timing pass)",
    }
    json.dump(out_j, open(os.path.join(here, f"traffic_{prec}.json"), "w"), indent=1)
    print("wrote", f"traffic_{prec}.json", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out_j.items() if k != "source"})
print("wrote", tag)
