#!/bin/bash
# registration_mode 3: the pre-registration's arg-max as a kernel of its own (TSD_PDF_ARGMAX_KERNEL=1) against the arg-max as the first
# workgroup of the registration's launch (k_icp_pre, the default in the fused scan); the mode-3 GPU tests first.  One gpurun call.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_tsdpdf.py tests/test_gpu_batch.py tests/test_gpu_facade.py -x -q 2>&1 | tail -2
python3 tools/fuzz_slam.py 80 9500 mode3 2>&1 | tail -1
run() { python3 bench.py --registration-mode 3 --steps 600 --warmup 20 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('$1: %.0f scans/s | %s' % (d['value'], {k: (round(1e3*v,2) if v else None) for k,v in s.items()}))"; }
for rep in 1 2 3; do
  TSD_PDF_ARGMAX_KERNEL=1 run "arg-max kernel        "
  run "arg-max in the k_icp launch"
done
