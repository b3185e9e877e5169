#!/bin/bash
# registration_mode 3: the GPU tests of the pre-registration, then an A/B of two versions of csrc/tsdpdf.hip in ONE gpurun call (the
# pool's boxes differ by several per cent), then the stamp build of k_pdf_prepare.
#   tools/mode3_ab.sh [baseline source beside tsdpdf.hip, e.g. tsdpdf_r5.hip = `git show <rev>:ohm_tsd_slam_amd/csrc/tsdpdf.hip`]
# variant 1 = the baseline (default: tsdpdf.hip itself), variant 2 = tsdpdf.hip
BASE=${1:-tsdpdf.hip}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m3
python -m pytest tests/test_gpu_tsdpdf.py tests/test_gpu_batch.py -x -q > gpurun_out/r6m3/tests.log 2>&1; tail -3 gpurun_out/r6m3/tests.log
DIAG_SRC=$BASE DIAG_DIR=diag_ab1 tools/diag_build.sh tsdpdf > /dev/null 2>&1 || echo "r5 variant failed to build"
DIAG_DIR=diag_ab2 tools/diag_build.sh tsdpdf > /dev/null 2>&1
DIAG_DIR=diag_st tools/diag_build.sh tsdpdf -DTSD_PDF_STAMPS > /dev/null 2>&1
for rep in 1 2 3; do for v in 2 1; do
TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_ab$v python3 bench.py --registration-mode 3 --steps 600 --warmup 20 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('variant $v: %.0f scans/s | %s' % (d['value'], {k: (round(1e3*v,2) if v is not None else None) for k,v in s.items()}))"
done; done
TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_st python3 bench.py --registration-mode 3 --steps 20 --warmup 2 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | grep -v "^{" | tail -12
