#!/bin/bash
# round 3: final evidence run -- full GPU suite, default bench, bench under rocprofv3 --kernel-trace --stats
O=gpurun_out/r03final
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 700 python3 -m pytest tests -q -m gpu --durations=8 > $O/gpu_tests.log 2>&1; echo tests rc=$?; tail -14 $O/gpu_tests.log
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
timeout -k 10 300 python3 bench.py --warmup 20 --steps 20 --no-cpu --no-cube > $O/bench_w20.json 2> $O/bench_w20.err; echo bench w20 rc=$?
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python3 $R/bench.py --no-cpu --no-cube > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err; echo rocprof rc=$?
cd $R
python3 tools/rocpd_top.py $O/prof/bench_results.db 40 > $O/bench_kernel_stats.txt
rm -f $O/prof/*.db
python3 -c "
import json
for f in ('bench_default','bench_w20','bench_under_rocprof'):
    d=json.loads(open('$O/%s.json'%f).read()); print(f, round(d['ms_per_step'],2), d['pcg_iterations_per_solve'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), d['setup_s']['pattern_and_upload'], round(d['north_star_workload']['analysis_s'],3))"
