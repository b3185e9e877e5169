#!/bin/bash
# per-kernel durations of the registration_mode 3 chain for library variants side by side (rocprofv3 --kernel-trace --stats):
#   tools/mode3_kernels.sh <lib dir under ohm_tsd_slam_amd/lib> ...        ("." = the product build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  d=gpurun_out/prof_m3_$(echo $v | tr './' '__')
  rm -rf $d
  TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o run -- python3 bench.py --registration-mode 3 --steps 300 --warmup 10 --no-cpu-baseline --no-second-pass --no-stream > $d.json 2> $d.err
  echo "== $v: $(python3 -c "import json;print(round(json.load(open('$d.json'))['value']))") scans/s under the profiler"
  python3 - $d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pdf" in r["Name"] or "icp" in r["Name"]:
        print("   %-60s calls %5s avg %8.2f us  min %8.2f  max %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  find $d -name "*kernel_trace.csv" -delete 2>/dev/null
done
