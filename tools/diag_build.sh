#!/bin/bash
# Builds an INSTRUMENTED / variant copy of the device library into ohm_tsd_slam_amd/lib/diag/ (never into lib/):
#   tools/diag_build.sh <translation unit without .hip> <extra hipcc flags...>
# One translation unit is recompiled with the flags, the others are the stock objects of lib/obj; the C++ facade is
# copied next to it (its rpath is $ORIGIN).  Run the tool with TSD_LIB_DIR=<repo>/ohm_tsd_slam_amd/lib/diag.
set -e
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tu=$1; shift
lib=$root/ohm_tsd_slam_amd/lib
diag=${DIAG_DIR:-diag}            # DIAG_DIR=diag_<name>: several variants side by side
mkdir -p $lib/$diag/obj
cd $root/ohm_tsd_slam_amd/csrc
extra=""; if [ "$tu" = "push_kernels" ]; then extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"; fi
src=${DIAG_SRC:-$tu.hip}          # DIAG_SRC=<other source of the same translation unit>: an A/B against a previous version kept beside it
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include $extra "$@" -c $src -o $lib/$diag/obj/$tu.o
objs=""
for o in $lib/obj/*.o; do b=$(basename $o); if [ "$b" = "$tu.o" ]; then objs="$objs $lib/$diag/obj/$tu.o"; else objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/$diag/libtsd_hip.so $objs
cp $lib/libohm_tsd_slam.so $lib/$diag/
echo "diag library: $lib/$diag (TSD_LIB_DIR)"
