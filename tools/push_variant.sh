#!/bin/bash
# diagnostic: rebuild push_kernels with -DTSD_UPDATE_BLOCK=$1 on the GPU box and run the bench
cd $GRAFT_REPO_ROOT/ohm_tsd_slam_amd/csrc
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include -DTSD_UPDATE_BLOCK=$1 -c push_kernels.hip -o ../lib/obj/push_kernels.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtsd_hip.so ../lib/obj/*.o
cd $GRAFT_REPO_ROOT && python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "push" 2>&1 | tail -1
python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('block $1', d['value'], d['ms_push_kernels'], d['roofline']['avg_launch_ms'])"
