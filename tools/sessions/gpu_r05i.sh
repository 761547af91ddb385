#!/bin/bash
# round 5, verdict item 3: the FP64 SpMV inside PCG with and without the HIP events around it, DOT variant, PCG-like x (kernel durations from a trace)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ev && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev -- python3 $GRAFT_REPO_ROOT/tools/sessions/spmv_events_r05.py > $O/events.out 2> $O/events.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
grep instance $O/events.out; tail -3 $O/events.err | cut -c1-300
python3 tools/sessions/spmv_events_r05.py --analyse /tmp/ev | tee $O/events.txt
