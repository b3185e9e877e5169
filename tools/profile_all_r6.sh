#!/bin/bash
# GPU box: every profile the round-6 numbers in DESIGN.md cite, summarised into gpurun_out/profiles_new/ (copied to profiles/ by hand).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/profiles_new
tools/profile_bench.sh r6_cfg2_pillars --occupancy 5
tools/profile_bench.sh r6_cfg2_comb_push --scene comb --mode push
tools/profile_bench.sh r6_cfg3_pillars_push --config cfg3 --scene pillars --mode push
tools/profile_bench.sh r6_cfg3_comb_push --config cfg3 --scene comb --mode push
tools/profile_sq.sh r6_cfg3_comb_push --config cfg3 --scene comb --mode push > gpurun_out/profiles_new/r6_cfg3_comb_push_sq_counters.txt 2>&1
tools/profile_sq.sh r6_cfg2_pillars > gpurun_out/profiles_new/r6_cfg2_pillars_sq_counters.txt 2>&1
# N3: registration_mode 3 (TSD_PDF pre-registration ahead of the ICP), kernel trace only
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6_cfg2_pillars_mode3_stats -o run -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream --registration-mode 3 > gpurun_out/prof_r6_cfg2_pillars_mode3_stats.json 2> gpurun_out/prof_r6_cfg2_pillars_mode3_stats.err
cp $(find gpurun_out/prof_r6_cfg2_pillars_mode3_stats -name "*kernel_stats.csv" | head -1) gpurun_out/profiles_new/r6_cfg2_pillars_mode3_kernel_stats.csv
cp gpurun_out/prof_r6_cfg2_pillars_mode3_stats.json gpurun_out/profiles_new/r6_cfg2_pillars_mode3_bench_under_rocprof.json
find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out -name "*counter_collection.csv" -delete 2>/dev/null
# un-profiled bench lines: the default (driver-like 20 steps and 200 steps), mode 3, 8 robots on one grid
timeout 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/profiles_new/r6_bench_20steps.json 2> gpurun_out/r6_bench20.err
timeout 300 python3 bench.py > gpurun_out/profiles_new/r6_bench_200steps.json 2> gpurun_out/r6_bench200.err
timeout 300 python3 bench.py --registration-mode 3 --no-cpu-baseline > gpurun_out/profiles_new/r6_bench_mode3.json 2> gpurun_out/r6_bench_mode3.err
timeout 300 python3 bench.py --async-mapping --no-cpu-baseline > gpurun_out/profiles_new/r6_bench_async_mapping.json 2> gpurun_out/r6_bench_async.err
# several robots on ONE grid (the reference's multi-robot mode): the curve, registration_mode 0 and 3
: > gpurun_out/profiles_new/r6_multi_robot_one_grid.json; : > gpurun_out/profiles_new/r6_multi_robot_mode3.json
for r in 1 2 4 8 12 16; do
  timeout 300 python3 bench.py --robots $r --no-cpu-baseline --no-second-pass --no-stream >> gpurun_out/profiles_new/r6_multi_robot_one_grid.json 2>> gpurun_out/r6_bench_robots.err
  timeout 300 python3 bench.py --robots $r --registration-mode 3 --no-cpu-baseline --no-second-pass --no-stream >> gpurun_out/profiles_new/r6_multi_robot_mode3.json 2>> gpurun_out/r6_bench_robots.err
done
# cfg 3 (16384^2) through the SLAM loop
timeout 600 python3 bench.py --config cfg3 --no-cpu-baseline > gpurun_out/profiles_new/r6_bench_cfg3_slam.json 2> gpurun_out/r6_bench_cfg3_slam.err
timeout 300 python3 bench.py --config cfg3 --scene comb --mode push --steps 100 --no-cpu-baseline > gpurun_out/profiles_new/r6_bench_cfg3_comb_push.json 2> gpurun_out/r6_bench_c3.err
timeout 300 python3 bench.py --gpus 1 --force-dist --no-cpu-baseline > gpurun_out/profiles_new/r6_bench_force_dist_1rank.json 2> gpurun_out/r6_bench_fd.err
ls -la gpurun_out/profiles_new
# the tail of the registration in the fused loop (every wave's stamps, the window-round histogram) and the mode-3 chain's kernels
tools/icp_tail.sh 220 --fused > gpurun_out/profiles_new/r6_icp_tail_fused.txt 2>&1
tools/icp_tail.sh 220 > gpurun_out/profiles_new/r6_icp_tail_unfused_hist.txt 2>&1
tools/mode3_kernels.sh . > gpurun_out/profiles_new/r6_mode3_kernels.txt 2>&1
ls -la gpurun_out/profiles_new
