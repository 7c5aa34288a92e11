#!/usr/bin/env python3
"""Merge one tools/reproduce.sh run into the machine-readable record bench.py checks its profile-sourced figures
against (profiles/counters.json):

  python tools/make_counters.py <tag>      reads  gpurun_out/prof_<tag>/{trace,pmc}/**, gpurun_out/traffic_<tag>.json,
                                                  gpurun_out/<tag>_bench.json (roofline.code_identity)
                                           writes gpurun_out/<tag>_counters.json

Per kernel of the sweep: SQ_INSTS_VALU / SALU per wave, SQ_WAIT_ANY / SQ_WAVE_CYCLES (separate rocprofv3 --pmc pass),
rocprofv3's average duration IN the sweep (kernel-trace pass of `bench.py --no-roofline`: no isolated timing loops
in that run), HBM bytes per launch (TCC passes, calibrated: tools/traffic.py), and `code_id` = the identity of the
code object the counters were taken on (hiprtc kernels: gmx_program_code_hash; AOT kernels: sha256 of the library).
bench.py uses a kernel's entry only when its code_id equals the running code's."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0].strip()
    if name.startswith("void "):
        name = name[5:]
    return name.split("<")[0]


def main(tag):
    out_dir = os.path.join(ROOT, "gpurun_out")
    bench = json.load(open(os.path.join(out_dir, f"{tag}_bench.json")))
    ids = bench["roofline"]["code_identity"]
    kernels = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(out_dir, f"prof_{tag}", "trace", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            kernels[k]["avg_us_in_sweep"] = float(r["AverageNs"]) / 1e3
            kernels[k]["calls_in_trace"] = int(r["Calls"])
            kernels[k]["full_name"] = r["Name"].split("(")[0][:80]
    # a background launch covers up to 10 steps (one 2-D launch per group and key): per STEP = total / chain launches
    if "gmx_jit_background_kernel" in kernels and "gmx_jit_kernel" in kernels:
        nb, nk = kernels["gmx_jit_background_kernel"], kernels["gmx_jit_kernel"]
        nb["avg_us_per_launch"] = nb["avg_us_in_sweep"]
        nb["avg_us_in_sweep"] = nb["avg_us_in_sweep"] * nb["calls_in_trace"] / nk["calls_in_trace"]
        nb["note"] = "avg_us_in_sweep is per STEP (total duration / site-program launches); one launch covers up to 10 steps"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, f"prof_{tag}", "pmc", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        c = {n: sum(x) / len(x) for n, x in v.items()}
        w = c.get("SQ_WAVES", 0) or 1
        kernels[k].update(waves=w, valu_per_wave=c.get("SQ_INSTS_VALU", 0) / w, salu_per_wave=c.get("SQ_INSTS_SALU", 0) / w,
                          lds_per_wave=c.get("SQ_INSTS_LDS", 0) / w,
                          wait_any_frac=(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None)
    tpath = os.path.join(out_dir, f"traffic_{tag}.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        for k, d in tj.get("kernels", {}).items():
            kernels[short(k)]["hbm_bytes_per_launch"] = d["hbm_bytes_corrected"]
    keep = {}
    for k, d in kernels.items():
        if k.startswith(("at::", "__amd", "rccl")) or "valu_per_wave" not in d:
            continue
        d["code_id"] = ids.get(k, ids.get("libgenmi_hip.so"))
        keep[k] = d
    res = {"tag": tag, "commit": os.environ.get("GENMI_COMMIT"), "workload": bench["config"]["workload"],
           "bench_value": bench["value"], "code_identity": ids, "kernels": keep}
    path = os.path.join(out_dir, f"{tag}_counters.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1)[:2500])


if __name__ == "__main__":
    main(sys.argv[1])
