#!/bin/bash
# round 5, verdict item 3: which ingredient of a PCG iteration makes the FP64 SpMV 15 % slower than back to back (kernel trace of phases)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -- python3 $GRAFT_REPO_ROOT/tools/sessions/spmv_phases_r05.py > $O/phases.out 2> $O/phases.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
grep solve $O/phases.out
python3 tools/sessions/spmv_phases_r05.py --analyse /tmp/ph | tee $O/phases.txt
