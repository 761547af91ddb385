#!/bin/bash
# round 3, call e: 16-bit block-scaled smoother copy: test, iteration counts, new bench.py, kernel trace
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "16bit_smoother" > $O/tests.log 2>&1
echo tests rc=$?; tail -5 $O/tests.log
V="add,lo,lo:env.FCVM_MG_K1S=3"
timeout -k 10 300 python3 tools/precond_probe.py --case plate --h 3.0 --variants "$V" --out $O/plate_h3.json > $O/plate_h3.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
timeout -k 10 300 python3 bench.py --plate-h 6 --steps 5 --warmup 2 --cells 10 > $O/bench_small.json 2> $O/bench_small.err; echo bench small rc=$?; tail -c 600 $O/bench_small.err
cd /tmp
FCVM_MG_ADDITIVE=2 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_lo -o lo -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-cube > $GRAFT_REPO_ROOT/$O/bench_lo.json 2> $GRAFT_REPO_ROOT/$O/bench_lo.err; echo bench lo rc=$?
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_add -o add -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-cube > $GRAFT_REPO_ROOT/$O/bench_add.json 2> $GRAFT_REPO_ROOT/$O/bench_add.err; echo bench add rc=$?
cd $GRAFT_REPO_ROOT
find $O -name "*.db" -size +20M -delete
ls -la $O $O/prof_lo 2>/dev/null | head -30
