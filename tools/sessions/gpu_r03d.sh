#!/bin/bash
# round 3, call d: FP16-smoother multiplicative cycle: test, iteration counts and times on whole analyses
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fp16_smoother or both_preconditioners or window_format" > $O/tests.log 2>&1
echo tests rc=$?; tail -5 $O/tests.log
S='env.FCVM_MG_MAX_AGGREGATES=100;env.FCVM_L2S_MAX_SUPER=1500'
V="add,lo,lo:env.FCVM_MG_K1S=3,lo:env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=4;env.FCVM_L2S_COARSE=1,lo:env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=2;env.FCVM_L2S_COARSE=1,lo:alpha0=3,lo:alpha0=8,mult1"
timeout -k 10 300 python3 tools/precond_probe.py --case plate --h 3.0 --variants "$V" --out $O/plate_h3.json > $O/plate_h3.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
tail -c 400 $O/plate_h3.log; tail -c 300 $O/plate_1m.log; tail -c 300 $O/emb3.log
