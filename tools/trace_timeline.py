#!/usr/bin/env python3
"""Print a window of a rocprofv3 --kernel-trace CSV as a timeline: per kernel its stream (queue), start relative to
the window, duration and the gap since the previous kernel ended — what a latency-bound chain is made of.
usage: tools/trace_timeline.py <dir with *_kernel_trace.csv> [--skip N] [--count M] [--anchor NAME]"""
import argparse
import csv
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip", type=int, default=0, help="occurrences of the anchor kernel to skip")
    ap.add_argument("--count", type=int, default=40)
    ap.add_argument("--anchor", default="k_shard_step")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel trace under " + a.dir)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48],
                             r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    seen, first = 0, None
    for i, r in enumerate(rows):
        if a.anchor in r[2]:
            if seen == a.skip:
                first = i
                break
            seen += 1
    if first is None:
        sys.exit("anchor not found")
    t0, prev_end = rows[first][0], None
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>8}  queue/stream  kernel")
    for s, e, name, q, st in rows[first:first + a.count]:
        gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.2f}"
        print(f"{(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:>8}  {q}/{st}  {name}")
        prev_end = max(e, prev_end or 0)


if __name__ == "__main__":
    main()
