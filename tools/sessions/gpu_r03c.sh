#!/bin/bash
# round 3, call c: four-level hierarchy (small aggregates, sparse level 2, dense super-aggregates) on whole analyses
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
S='env.FCVM_MG_MAX_AGGREGATES=100;env.FCVM_L2S_MAX_SUPER=1500'
V="add,add:env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=2;env.FCVM_L2S_COARSE=1,add:env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=4;env.FCVM_L2S_COARSE=1,add:env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=8;env.FCVM_L2S_COARSE=2,add:env.FCVM_MG_TARGET=10;$S;env.FCVM_L2S_K=4;env.FCVM_L2S_COARSE=1,add:env.FCVM_MG_TARGET=48;$S;env.FCVM_L2S_K=4;env.FCVM_L2S_COARSE=1,mult1:env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=4;env.FCVM_L2S_COARSE=1,mult1:env.FCVM_MG_TARGET=24;$S;env.FCVM_L2S_K=8;env.FCVM_L2S_COARSE=2;env.FCVM_MG_K1S=3"
FCVM_VERBOSE=1 timeout -k 10 300 python3 tools/precond_probe.py --case plate --h 3.0 --variants "$V" --out $O/plate_h3.json > $O/plate_h3.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
tail -c 400 $O/plate_h3.log; tail -c 300 $O/plate_1m.log; tail -c 300 $O/emb3.log
