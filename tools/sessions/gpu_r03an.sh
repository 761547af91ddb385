#!/bin/bash
# round 3: aggregate size of the dense aggregate level revisited now that its inverse is cheaper (whole analyses)
O=gpurun_out/r03an
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo,lo:env.FCVM_MG_TARGET=128,lo:env.FCVM_MG_TARGET=100,lo:env.FCVM_MG_TARGET=80,lo:env.FCVM_MG_TARGET=64,lo:env.FCVM_MG_TARGET=100;env.FCVM_MG_K1S=4,lo:env.FCVM_MG_TARGET=100;env.FCVM_MG_K1S=2"
timeout -k 10 500 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.txt 2>&1 || exit 1
cut -c1-400 $O/plate_1m.txt | grep variant
timeout -k 10 400 python3 tools/precond_probe.py --case embankment --refine 3 --variants "lo,lo:env.FCVM_MG_TARGET=100,lo:env.FCVM_MG_TARGET=64" --out $O/emb3.json > $O/emb3.txt 2>&1 || exit 1
cut -c1-400 $O/emb3.txt | grep variant
