#!/bin/bash
O=gpurun_out/r03w
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo,lo:alpha0=8,lo:alpha0=16,lo:alpha0=40,lo:w=0.9,lo:alpha0=16;env.FCVM_MG_ALPHA1S=20,lo:alpha0=16;env.FCVM_MG_ALPHA1S=8"
timeout -k 10 900 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
