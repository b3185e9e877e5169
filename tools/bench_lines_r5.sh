#!/bin/bash
# GPU box: the un-profiled bench lines of profiles/r5_* once more (the tail of tools/profile_all_r5.sh without the rocprofv3 passes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/profiles_new
timeout 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/profiles_new/r5_bench_20steps.json 2> gpurun_out/r5_bench20.err
timeout 300 python3 bench.py > gpurun_out/profiles_new/r5_bench_200steps.json 2> gpurun_out/r5_bench200.err
timeout 300 python3 bench.py --registration-mode 3 --no-cpu-baseline > gpurun_out/profiles_new/r5_bench_mode3.json 2> gpurun_out/r5_bench_mode3.err
timeout 300 python3 bench.py --async-mapping --no-cpu-baseline > gpurun_out/profiles_new/r5_bench_async_mapping.json 2> gpurun_out/r5_bench_async.err
: > gpurun_out/profiles_new/r5_multi_robot_one_grid.json; : > gpurun_out/profiles_new/r5_multi_robot_mode3.json
for r in 1 2 4 8 12 16; do
  timeout 300 python3 bench.py --robots $r --no-cpu-baseline --no-second-pass --no-stream >> gpurun_out/profiles_new/r5_multi_robot_one_grid.json 2>> gpurun_out/r5_bench_robots.err
  timeout 300 python3 bench.py --robots $r --registration-mode 3 --no-cpu-baseline --no-second-pass --no-stream >> gpurun_out/profiles_new/r5_multi_robot_mode3.json 2>> gpurun_out/r5_bench_robots.err
done
timeout 600 python3 bench.py --config cfg3 --no-cpu-baseline > gpurun_out/profiles_new/r5_bench_cfg3_slam.json 2> gpurun_out/r5_bench_cfg3_slam.err
