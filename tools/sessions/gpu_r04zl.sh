#!/bin/bash
# round 4: kernel timeline of one Newton iteration of the bench, the part outside the PCG iterations
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/nt && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/nt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-cube > $GRAFT_REPO_ROOT/gpurun_out/r04zl_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04zl_bench.err
cd $GRAFT_REPO_ROOT
python3 tools/sessions/newton_timeline_r04.py /tmp/nt 30 > gpurun_out/r04zl_timeline.txt
cat gpurun_out/r04zl_timeline.txt
