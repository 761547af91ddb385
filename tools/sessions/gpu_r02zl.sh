#!/bin/bash
O=gpurun_out/r02zl
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "not elastica and not bench_line" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-plate > $GRAFT_REPO_ROOT/$O/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-60,220-300
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*' $O/bench_trace.log
timeout -k 10 300 python3 tools/example_run.py plate --h 1.45 > $O/plate.log 2>&1; tail -1 $O/plate.log | cut -c1-260
