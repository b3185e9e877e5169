#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats, then HBM traffic counters in separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit in one pass on gfx950).
#   usage: tools/profile_bench.sh <tag> [bench args]      e.g.  tools/profile_bench.sh r2_cfg3_comb_push --config cfg3 --scene comb
# <tag> = <round>_<workload key of bench.py> (cfg_scene[_push][_q32]): bench.py reads profiles/<tag>_pmc.json for
# `roofline.traffic`.  Output under gpurun_out/prof_<tag>_{stats,fetch,write}; tools/profile_summarise.py <tag> turns
# them into the small files committed under profiles/.
set -u
tag=$1; shift
export TMPDIR=/tmp
args="--steps 100 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream --calibrate $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -o run -- python3 bench.py $args > gpurun_out/prof_${tag}_stats.json 2> gpurun_out/prof_${tag}_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -o run -- python3 bench.py $args > gpurun_out/prof_${tag}_fetch.json 2> gpurun_out/prof_${tag}_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -o run -- python3 bench.py $args > gpurun_out/prof_${tag}_write.json 2> gpurun_out/prof_${tag}_write.err
# keep what travels back small: the per-dispatch traces are not needed once summarised
python3 tools/profile_summarise.py ${tag} > gpurun_out/prof_${tag}_summary.txt 2>&1
find gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write -name "*kernel_trace.csv" -delete 2>/dev/null
find gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write -name "*counter_collection.csv" -delete 2>/dev/null
tail -5 gpurun_out/prof_${tag}_summary.txt
