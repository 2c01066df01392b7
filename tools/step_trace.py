#!/usr/bin/env python3
"""Kernel sequence of ONE training step out of a rocprofv3 --kernel-trace CSV: the launches between two consecutive
adamw_kernel launches (the last full step of the trace), in start order, with durations and the idle gap in front of each.
  python tools/step_trace.py <dir with *_kernel_trace.csv> [out.md] [steps back from the last one, default 0]
(bench.py's last block runs launch by launch with HIP events: with --steps K pass K to look at a step of the timed blocks)"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"gts::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        raise SystemExit("no *kernel_trace.csv under " + sys.argv[1])
    rows = []
    with open(paths[0]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    if len(ends) < 2:
        raise SystemExit("fewer than two optimizer launches in the trace")
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if len(ends) < back + 2:
        raise SystemExit("not that many steps in the trace")
    a, b = ends[-2 - back] + 1, ends[-1 - back] + 1
    step = rows[a:b]
    out = ["| # | kernel | us | gap us |", "|---:|---|---:|---:|"]
    prev_end = rows[a - 1][1]
    busy = 0
    for i, (s, e, n) in enumerate(step):
        out.append(f"| {i} | `{short(n)}` | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} |")
        busy += e - s
        prev_end = max(prev_end, e)
    wall = step[-1][1] - rows[a - 1][1]
    out.append("")
    out.append(f"{len(step)} launches, kernel time {busy / 1e3:.1f} us, wall {wall / 1e3:.1f} us (from the previous step's last kernel)")
    text = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
