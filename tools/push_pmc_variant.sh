#!/bin/bash
# diagnostic: WRITE_SIZE / FETCH_SIZE of k_push_update for a build variant ($TSD_EXTRA, built into lib/diag), push-only bench
# usage: TSD_EXTRA="-DTSD_RMW_RECORDS" tools/push_pmc_variant.sh [bench args]
$GRAFT_REPO_ROOT/tools/diag_build.sh push_kernels $TSD_EXTRA
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf gpurun_out/prof_variant
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_variant -o run -- python3 bench.py --mode push --steps 60 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/prof_variant.err
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_variant/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_push_update" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant [$TSD_EXTRA]", {k: round(sum(v) / len(v), 1) for k, v in acc.items()}, "KB per launch")
PY
done
rm -rf gpurun_out/prof_variant
