#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs (kernel trace + PMC passes) into a small table."""
import collections
import csv
import glob
import sys


def main(root):
    for f in sorted(glob.glob(f"{root}/**/*_kernel_stats.csv", recursive=True)):
        print("== kernel stats", f)
        for r in csv.DictReader(open(f)):
            if float(r["Percentage"]) < 0.5:
                continue
            print(f'{r["Name"][:60]:60s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"]) / 1e3:8.2f} '
                  f'min_us={float(r["MinNs"]) / 1e3:7.2f} pct={r["Percentage"]}')
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    waves = {}
    for f in sorted(glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][:40]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith(("void at::", "__amd")):
            continue
        c = {n: sum(x) / len(x) for n, x in v.items()}
        w = c.get("SQ_WAVES", 0) or 1
        print(f"== pmc {k}: waves={w:.0f}")
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
            if n in c:
                print(f"   {n:22s} per wave {c[n] / w:9.1f}")
        for n in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if n in c:
                print(f"   {n:22s} cycles per wave {4 * c[n] / w:9.0f}")
        if "GRBM_GUI_ACTIVE" in c:
            print(f"   GRBM_GUI_ACTIVE/8 = {c['GRBM_GUI_ACTIVE'] / 8:.0f} cycles")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
