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


def _git_head():
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        return None


def other_configs(tag, out_dir, ids):
    """BASELINE configs 3 / 4 / 5 (tools/prof_config.sh): the process's total SQ_INSTS_VALU (wave instructions) per
    workload unit, per kernel: instructions per wave, waves per launch, launches per unit, rocprofv3's average duration;
    the kernel with the most GPU time is the dominant one.  `lib` = sha256 of the library the counters were taken on."""
    res = {}
    for k in (3, 4, 5):
        root = os.path.join(out_dir, f"prof_{tag}_config{k}")
        upath = os.path.join(out_dir, f"{tag}_config{k}_units_pmc.json")
        if not os.path.isdir(root) or not os.path.exists(upath):
            continue
        try:
            units = json.loads(open(upath).read().strip().splitlines()[-1])
        except Exception:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(root, "pmc", "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        stats = {}
        for f in glob.glob(os.path.join(root, "trace", "**", "*_kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                stats[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]), int(r["Calls"]))
        kernels, total = {}, 0.0
        for name, c in per.items():
            if "SQ_INSTS_VALU" not in c or name.startswith(("at::", "__amd")):
                continue
            valu, waves = sum(c["SQ_INSTS_VALU"]), sum(c.get("SQ_WAVES", [0]))
            if valu <= 0:
                continue
            total += valu
            kernels[name] = {"launches_per_unit": len(c["SQ_INSTS_VALU"]) / units["units"],
                             "waves_per_launch": waves / max(1, len(c["SQ_INSTS_VALU"])),
                             "valu_insts_per_wave": valu / max(1.0, waves),
                             "avg_us_in_profile": stats.get(name, (None,))[0]}
        if not kernels:
            continue
        dom = max(kernels, key=lambda n_: stats.get(n_, (0, 0.0, 0))[1])
        res[f"config{k}"] = {"tag": tag, "lib": ids.get("libgenmi_hip.so"), "programs": units.get("programs"), "unit": units["unit"], "units_profiled": units["units"],
                             "valu_wave_insts_per_unit": total / units["units"], "kernels": kernels, "dominant_kernel": dom,
                             "dominant_valu_per_wave": kernels[dom]["valu_insts_per_wave"],
                             "dominant_avg_us": kernels[dom]["avg_us_in_profile"]}
    return res


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
    res = {"tag": tag, "commit": os.environ.get("GENMI_COMMIT") or _git_head(), "workload": bench["config"]["workload"],
           "bench_value": bench["value"], "code_identity": ids, "kernels": keep, "configs": other_configs(tag, out_dir, ids)}
    path = os.path.join(out_dir, f"{tag}_counters.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1)[:2500])


if __name__ == "__main__":
    main(sys.argv[1])
