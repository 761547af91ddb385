#!/bin/bash
# round 4: kernel timeline of one tangent refresh of the GNLY probe
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python3 $GRAFT_REPO_ROOT/tools/gnly_probe.py --steps 6 > $GRAFT_REPO_ROOT/gpurun_out/r04zi_run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/sessions/gnly_timeline_r04.py /tmp/gt 3 > gpurun_out/r04zi_timeline.txt
grep "^iteration" gpurun_out/r04zi_run.log | tail -3 | cut -c1-150
cat gpurun_out/r04zi_timeline.txt
