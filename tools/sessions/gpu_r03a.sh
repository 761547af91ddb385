#!/bin/bash
# round 3, call a: solver-space probe (Ritz spectrum of the V-cycle-preconditioned operator, init / deflated PCG iteration counts)
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_VERBOSE=1 timeout -k 10 300 python3 tools/deflate_probe.py --case plate --h 3.0 --variants base,init16,defl16,defl32 --out $O/plate_h3.json > $O/plate_h3.log 2>&1 &&
timeout -k 10 500 python3 tools/deflate_probe.py --case plate --h 1.45 --variants base,init32,defl16,defl32 --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 500 python3 tools/deflate_probe.py --case embankment --refine 3 --variants base,defl16,defl32 --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
tail -c 600 $O/plate_h3.log; tail -c 300 $O/plate_1m.log; tail -c 300 $O/emb3.log
