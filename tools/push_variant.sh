#!/bin/bash
# diagnostic: rebuild push_kernels with extra -D flags ($TSD_EXTRA) on the GPU box and run the bench (no parity!)
cd $GRAFT_REPO_ROOT/ohm_tsd_slam_amd/csrc
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include $TSD_EXTRA -c push_kernels.hip -o ../lib/obj/push_kernels.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtsd_hip.so ../lib/obj/*.o
cd $GRAFT_REPO_ROOT && python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$TSD_EXTRA]', d['value'], 'push kernels', d['ms_push_kernels'], 'update', d['roofline']['avg_launch_ms'])"
