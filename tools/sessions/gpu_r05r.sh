#!/bin/bash
# round 5: cost / benefit of the replicated aggregate level's settings at the 6-rank size -- ONE rank holding the 6.0 M mesh through the host-sequenced layer
# with the aggregates a 6-rank run gets (FCVM_MG_TARGET=69 -> 13 k aggregates, sparse level): smoothing steps per sweep (FCVM_L2S_K) and super-aggregate
# corrections (FCVM_L2S_COARSE) against PCG iterations and time per iteration; and the dense level at 3000 aggregates (FCVM_MG_TARGET=300)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O
cd $R
for v in ${SCAN:-"69 8 2" "69 4 2" "69 4 1" "69 2 2" "69 2 1" "300 8 2"}; do
  set -- $(echo $v | tr "," " ")
  echo "== FCVM_MG_TARGET=$1 FCVM_L2S_K=$2 FCVM_L2S_COARSE=$3" | tee -a $O/scan.txt
  PLATE_H=0.79796 FORCE_MODES=1 FCVM_VERBOSE=1 FCVM_MG_TARGET=$1 FCVM_L2S_K=$2 FCVM_L2S_COARSE=$3 timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py 2>&1 | grep -E "host-sequenced|fcvm dist: aggregate" | tail -2 | cut -c1-260 | tee -a $O/scan.txt
done
