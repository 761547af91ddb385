#!/bin/bash
O=gpurun_out/r03h
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo:env.FCVM_MG_K1S=3;env.FCVM_RHS_PROJECTION=0,lo:env.FCVM_MG_K1S=3;env.FCVM_RHS_PROJECTION=3,lo:env.FCVM_MG_K1S=3;env.FCVM_RHS_PROJECTION=6,lo:env.FCVM_MG_K1S=3;env.FCVM_RHS_PROJECTION=12,lo:env.FCVM_MG_K1S=3;env.FCVM_RHS_PROJECTION=20,lo:env.FCVM_MG_K1S=3;env.FCVM_RHS_PROJECTION=40"
timeout -k 10 900 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
