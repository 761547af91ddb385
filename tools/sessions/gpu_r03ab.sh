#!/bin/bash
# Ritz spectrum and deflation with the multiplicative cycle (form 2): are there isolated low eigenvalues now?
O=gpurun_out/r03ab
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python3 tools/deflate_probe.py --case plate --h 1.45 --variants base,init16,defl8,defl24 --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 500 python3 tools/deflate_probe.py --case embankment --refine 3 --variants base,defl16 --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
