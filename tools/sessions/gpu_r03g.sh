#!/bin/bash
O=gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
S='env.FCVM_MG_MAX_AGGREGATES=100;env.FCVM_L2S_MAX_SUPER=1500'
V="lo,lo:env.FCVM_MG_K1S=3,lo:env.FCVM_MG_K1S=4,lo:env.FCVM_MG_K1S=5,lo:env.FCVM_MG_K1S=6,lo:env.FCVM_MG_K1S=4;env.FCVM_MG_ALPHA1S=8,lo:env.FCVM_MG_K1S=4;env.FCVM_MG_ALPHA1S=30,lo:env.FCVM_MG_K1S=3;env.FCVM_MG_TARGET=100,lo:env.FCVM_MG_K1S=2;env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=4;env.FCVM_L2S_COARSE=1,lo:env.FCVM_MG_K1S=3;env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=4;env.FCVM_L2S_COARSE=1,lo:env.FCVM_MG_K1S=3;env.FCVM_MG_TARGET=40;$S;env.FCVM_L2S_K=3;env.FCVM_L2S_COARSE=1"
timeout -k 10 900 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
