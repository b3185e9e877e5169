#!/bin/bash
# The push's halo pass as a kernel of its own (TSD_HALO_KERNEL=1: round 5's form) against the pass in the prologue of the ray cast that
# follows the push (the default), and that with the cells read by plain loads behind the wait (-DTSD_RC_HALO_PLAIN, A/B only); one call.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_facade.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_async_mapping.py -x -q 2>&1 | tail -3
DIAG_DIR=diag_plain tools/diag_build.sh raycast_kernels -DTSD_RC_HALO_PLAIN > /dev/null 2>&1 || echo "plain variant failed to build"
run() { python3 bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('$1: %.0f scans/s | %s' % (d['value'], {k: (round(1e3*v,2) if v else None) for k,v in s.items()}))"; }
for rep in 1 2 3; do
  TSD_HALO_KERNEL=1 run "halo kernel        "
  run "halo in the ray cast"
  TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_plain run "  ... plain loads   "
done
