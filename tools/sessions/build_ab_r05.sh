#!/bin/bash
# A/B builds of the library for the slow-box experiment of round 5 (tools/sessions/gpu_r05d.sh): built HERE (hipcc cross-compiles), they
# travel to the GPU box with the snapshot (*.so is git-ignored, not gpurun-ignored).  Timing experiments: all but minwaves3 give wrong results.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/sessions/ab
for v in "notr:-DFCVM_AB_NO_TR" "minwaves3:-DFCVM_SYM_MINWAVES=3" "plainlds:-DFCVM_AB_PLAIN_LDS" "nofixed:-DFCVM_AB_NO_FIXED"; do
  name=${v%%:*}; flag=${v#*:}
  make -C fcvm-workbench_amd/csrc OUT=../../tools/sessions/ab/libfcvm_$name.so EXTRA="$flag" > /dev/null
  echo "built tools/sessions/ab/libfcvm_$name.so ($flag)"
done
