#!/bin/bash
# round 4: timeline of a pair iteration with the side stream
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 $GRAFT_REPO_ROOT/tools/sessions/pair_time_r04.py > $GRAFT_REPO_ROOT/gpurun_out/r04y_run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/sessions/pair_timeline_r04.py /tmp/pt 60 > gpurun_out/r04y_timeline.txt
tail -3 gpurun_out/r04y_run.log
