#!/bin/bash
# round 5: the super-aggregate level's inverse from its lower triangle (k_symv_tiles) against the full FP32 product (k_gemv_f32x4) at the
# 6-rank aggregate count (one rank, 6.0 M mesh, FCVM_MG_TARGET=69), plus the tests that cover the sparse level
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O; rm -f $O/scan.txt
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -k "sparse" -x 2>&1 | tail -3 | tee $O/tests.txt &&
for sy in 1 0; do
  echo "== FCVM_L2_SYMV=$sy FCVM_L2S_MAX_SUPER=2000 (FCVM_MG_TARGET=69)" | tee -a $O/scan.txt
  PLATE_H=0.79796 FORCE_MODES=1 FCVM_VERBOSE=1 FCVM_MG_TARGET=69 FCVM_L2_SYMV=$sy timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py 2>&1 | grep -E "host-sequenced|fcvm dist: aggregate" | tail -2 | cut -c1-300 | tee -a $O/scan.txt
done
