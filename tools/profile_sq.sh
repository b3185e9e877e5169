#!/bin/bash
# SQ counters of every kernel of the bench (one PMC pass): where the wave cycles go.  usage: tools/profile_sq.sh <tag>
set -u
tag=$1; shift
export TMPDIR=/tmp
ctrs=${TSD_SQ_COUNTERS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES}
timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/prof_${tag}_sq -o run -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream "$@" > gpurun_out/prof_${tag}_sq.json 2> gpurun_out/prof_${tag}_sq.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_${tag}_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[n]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, d in acc.items():
    print(n, {k: round(sum(v) / len(v), 1) for k, v in d.items()})
PY
