#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats, then HBM traffic counters in separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit in one pass on gfx950).  usage: tools/profile_bench.sh <tag> [bench args]
# Output under gpurun_out/prof_<tag>_{stats,fetch,write}; summaries are copied into profiles/ by hand.
set -u
tag=$1; shift
export TMPDIR=/tmp
args="--steps 100 --warmup 5 --no-cpu-baseline --calibrate $*"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -o run -- python3 bench.py $args > gpurun_out/prof_${tag}_stats.json 2> gpurun_out/prof_${tag}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -o run -- python3 bench.py $args > gpurun_out/prof_${tag}_fetch.json 2> gpurun_out/prof_${tag}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -o run -- python3 bench.py $args > gpurun_out/prof_${tag}_write.json 2> gpurun_out/prof_${tag}_write.err
ls gpurun_out/prof_${tag}_*/ 
