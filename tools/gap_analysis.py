#!/usr/bin/env python3
"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: per scan (k_raycast to the next k_raycast) how much of the
period is kernel time and where the gaps are.  usage: gap_analysis.py <dir>"""
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("tsd::", "")))
mc = glob.glob(os.path.join(sys.argv[1], "**", "*memory_copy_trace.csv"), recursive=True)
for m in mc:
    for r in csv.DictReader(open(m)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy_" + r.get("Direction", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_raycast")]
starts = starts[len(starts) // 2:]          # steady state
per = collections.defaultdict(list)
periods = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b + 1]
    periods.append((seg[-1][0] - seg[0][0]) / 1e3)
    for x, y in zip(seg[:-1], seg[1:]):
        per[f"{x[2][:18]:18s} -> {y[2][:18]:18s}"].append((y[0] - x[1]) / 1e3)
        per["run " + x[2]].append((x[1] - x[0]) / 1e3)
print("scans analysed", len(periods), "period avg %.1f us" % (sum(periods) / len(periods)))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:50s} n/scan {len(v) / len(periods):4.1f}  avg {sum(v) / len(v):8.2f} us  per scan {sum(v) / len(periods):8.2f} us")
