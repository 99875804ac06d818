#!/usr/bin/env python3
"""python profiles/summarize_variants.py gpurun_out r02_f16_variants  ->  profiles/<tag>.csv
Per timing variant of the fp16 kernel (profiles/collect_variants.sh): kernel time without counters, and from the PMC pass the
clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), the MFMA-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)), their
product (= MFMA-busy cycles per nanosecond, what the time must follow if the kernel is pipe-bound) and the wait fractions."""
import csv, glob, os, sys
src, tag = sys.argv[1], sys.argv[2]
kname = sys.argv[3] if len(sys.argv) > 3 else "nerf_mlp_f16_kernel"          # kernel whose dispatches are summarised
names = sys.argv[4].split(",") if len(sys.argv) > 4 else None                # variant directories to include (default: all)
here = os.path.dirname(os.path.abspath(__file__))
newest = lambda p: max(glob.glob(p), key=os.path.getmtime)
rows = []
for d in sorted(glob.glob(os.path.join(src, "var_*"))):
    name = os.path.basename(d)[4:]
    if names is not None and name not in names:
        continue
    st = [r for r in csv.DictReader(open(newest(os.path.join(d, "trace", "*", "*_kernel_stats.csv")))) if kname in r["Name"]]
    c, ns = {}, 0.0
    for r in csv.DictReader(open(newest(os.path.join(d, "pmc", "*", "*_counter_collection.csv")))):
        if kname not in r["Kernel_Name"]:
            continue
        c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ns += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    g, wc = c["GRBM_GUI_ACTIVE"], c["SQ_WAVE_CYCLES"]
    clk = g / ns / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 128.0)
    rows.append([name, f"{float(st[0]['AverageNs']) / 1e6:.3f}", f"{clk:.3f}", f"{busy:.4f}", f"{clk * busy:.4f}",
                 f"{c['SQ_WAIT_ANY'] / wc:.4f}", f"{c['SQ_WAIT_INST_ANY'] / wc:.4f}", f"{c['SQ_ACTIVE_INST_VALU'] / wc:.4f}",
                 f"{c['SQ_LDS_BANK_CONFLICT'] / max(1.0, c['SQ_BUSY_CYCLES']):.5f}"])
with open(os.path.join(here, tag + ".csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["variant", "kernel_ms_no_counters", "clock_ghz_pmc_pass", "mfma_busy_frac", "busy_x_clock", "wave_wait_any_frac",
                "wave_wait_inst_any_frac", "valu_inst_per_wave_cycle", "lds_bank_conflict_per_busy_cycle"])
    w.writerows(rows)
for r in rows:
    print(r)
