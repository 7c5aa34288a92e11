#!/usr/bin/env python3
"""Per-launch HBM traffic of the sweep's kernels from rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE, each in its own pass, never combined with tracing).

MI355X_MICROARCH.md §HBM: the counters are in KiB; on gfx950 FETCH_SIZE counts
exactly half the bytes of a 16-B/lane coalesced stream, other widths are
uncalibrated — so the dword-per-lane pattern these kernels use is calibrated here
on copy kernels of known size (tools/calib: 4 MB and 256 MB, dword and dwordx4).
Writes profiles/traffic.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def main(bench_root, calib_root, out):
    b, c = per_kernel(bench_root), per_kernel(calib_root)
    # calibration: tools/calib runs k_copy1 (dword/lane) and k_copy4 (dwordx4/lane) on 4 MB x201 and
    # 256 MB x21 launches; every launch reads as many bytes as it writes, and WRITE_SIZE is exact for
    # streaming stores, so the read-side correction factor is WRITE_SIZE / FETCH_SIZE.
    corr = {}
    for name, d in c.items():
        if name in ("k_copy1", "k_copy4") and d.get("FETCH_SIZE"):
            corr[name] = d["WRITE_SIZE"] / d["FETCH_SIZE"]
    f_rd = corr.get("k_copy1", 2.0)
    res = {"unit": "bytes per launch (1e6 particles)", "fetch_correction_factor": corr,
           "note": "FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts half the streamed bytes "
                   "(MI355X_MICROARCH.md §HBM), confirmed here for dword/lane and dwordx4/lane copies",
           "kernels": {}}
    for k, d in b.items():
        if k.startswith(("void at", "__amd")):
            continue
        fetch, write = d.get("FETCH_SIZE", 0) * 1024, d.get("WRITE_SIZE", 0) * 1024
        res["kernels"][k] = {"FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
                             "hbm_bytes_corrected": f_rd * fetch + write}
        if k.startswith("gmx_jit_kernel") or k.startswith("void k_vm") or k.startswith("k_vm_lean"):
            res["k_vm_hbm_bytes_per_launch"] = f_rd * fetch + write
        if k.startswith("gmx_jit_background_kernel"):
            res["noise_hbm_bytes_per_launch"] = f_rd * fetch + write
        if k.startswith("void k_offspring_tile"):
            res["k_offspring_tile_hbm_bytes_per_launch"] = f_rd * fetch + write
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
