#!/bin/bash
# round 5, verdict item 3: does a near-idle gap in front of the FP64 SpMV slow it down (kernel durations from a rocprofv3 trace)?
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python3 $GRAFT_REPO_ROOT/tools/sessions/spmv_gaps_r05.py > $O/gaps.out 2> $O/gaps.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
grep instance $O/gaps.out
python3 tools/sessions/spmv_gaps_r05.py --analyse /tmp/gp | tee $O/gaps.txt
