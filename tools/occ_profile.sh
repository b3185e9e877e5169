#!/bin/bash
# GPU box: the occupancy tests, then k_occ_cells / k_occ_mark under rocprofv3 (bench.py --occupancy 5 on the map of 40 scans)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4i
timeout 600 python -m pytest tests -x -q -m gpu -k "occup or color or merge or nranks or map_extraction" 2>&1 | tail -3
rm -rf gpurun_out/r4i/occx
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4i/occx -o run -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream --occupancy 5 > gpurun_out/r4i/occx.json 2> gpurun_out/r4i/occx.err
python3 - <<EOF
import csv,glob
f=glob.glob("gpurun_out/r4i/occx/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.reader(open(f)):
    if "occ" in r[0]: print(r[0][:30], r[1:4])
EOF
rm -rf gpurun_out/r4i/occx
