#!/bin/bash
# diagnostic (GPU box): where k_push_update's vector instructions go.  Builds the kernel with parts compiled OUT (-DTSD_ABLATE=<bits>:
# 1 no fix-up, 2 no exact part, 4 no increaseEmptiness tiles, 8 classification only) -- the results of those builds are wrong on
# purpose -- and reads SQ_INSTS_VALU / the duration of k_push_update per launch from one PMC pass each.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ab in ${TSD_ABLATE_SET:-0 1 2 4 6 14} "$@"; do
  DIAG_DIR=diag_ab$ab tools/diag_build.sh push_kernels -DTSD_ABLATE=$ab $TSD_ABLATE_EXTRA > /dev/null 2>&1 || { echo "ablate $ab failed to build"; continue; }
  export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_ab$ab
  for w in "cfg3 comb" "cfg2 pillars"; do set -- $w
    rm -rf gpurun_out/ab_tmp
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/ab_tmp -o run -- python3 bench.py --config $1 --scene $2 --mode push --steps 30 --warmup 3 --no-cpu-baseline --no-stream > /dev/null 2>&1
    python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/ab_tmp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_push_update" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("ablate $ab $1/$2:", {k: round(sum(v[3:]) / max(1, len(v[3:])), 1) for k, v in acc.items()})
PY
  done
  unset TSD_LIB_DIR
done
