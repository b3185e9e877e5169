#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv (+ kernel_trace.csv) per kernel: average duration
and average counter values per dispatch.  usage: pmc_summary.py <dir containing *_counter_collection.csv>"""
import collections, csv, glob, os, sys
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"].split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[name]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
for name, c in sorted(acc.items()):
    n = max(len(v) for k, v in c.items() if k != "_dur_ns")
    print(f"{name}: dispatches {n}, avg_us {sum(c['_dur_ns']) / len(c['_dur_ns']) / 1e3:.1f}")
    for k, v in sorted(c.items()):
        if k != "_dur_ns":
            print(f"    {k:42s} {sum(v) / len(v):14.1f}")
