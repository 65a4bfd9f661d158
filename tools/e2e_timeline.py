#!/usr/bin/env python3
"""Timeline of one search level of `bench.py --config e2e` from a rocprofv3 trace directory:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 bench.py --config e2e --keyframes 20
    python3 tools/e2e_timeline.py DIR
Prints, for the steady-state levels, each operation's mean duration and the mean idle gap before it."""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
ops = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ops.sort()
# a level ends with the grid kernel; take the last 30 levels
ends = [i for i, o in enumerate(ops) if "nmi_grid_kernel" in o[2]]
levels = []
for a, b in zip(ends[-31:-1], ends[-30:]):
    levels.append(ops[a + 1:b + 1])
n = min(len(l) for l in levels)
levels = [l for l in levels if len(l) == n] if len(set(len(l) for l in levels)) > 1 else levels
print(f"{len(levels)} levels, {len(levels[0])} operations each")
prev_end = np.array([ops[ends[-31 + i]][1] for i in range(len(levels))], dtype=np.float64) if len(levels) == 30 else None
tot = 0.0
for k in range(len(levels[0])):
    dur = np.mean([l[k][1] - l[k][0] for l in levels]) / 1e3
    if k == 0:
        gap = float("nan")
    else:
        gap = np.mean([l[k][0] - l[k - 1][1] for l in levels]) / 1e3
    print(f"  {levels[0][k][2]:50s} {dur:8.1f} us   gap before {gap:7.1f} us")
span = np.mean([l[-1][1] - l[0][0] for l in levels]) / 1e3
period = np.mean(np.diff([l[-1][1] for l in levels])) / 1e3
print(f"first start -> grid end: {span:.1f} us; level period (grid end to grid end): {period:.1f} us")
