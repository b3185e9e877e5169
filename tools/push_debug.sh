#!/bin/bash
cd $GRAFT_REPO_ROOT/ohm_tsd_slam_amd/csrc
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include -DTSD_PUSH_DEBUG -c push_kernels.hip -o ../lib/obj/push_kernels.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtsd_hip.so ../lib/obj/*.o
cd $GRAFT_REPO_ROOT && python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep -c "unsure lanes"
python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "unsure lanes" | head -20
