#!/bin/bash
# round 3, call b: cycle settings on whole analyses (where does the dense low end of the preconditioned spectrum come from?)
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V='add,mult1,mult2,mult3,add:env.FCVM_MG_K1S=4,add:env.FCVM_MG_TARGET=64,add:w=0.6,add:w=1.6,mult1:env.FCVM_MG_K1S=4'
timeout -k 10 300 python3 tools/precond_probe.py --case plate --h 3.0 --variants "$V" --out $O/plate_h3.json > $O/plate_h3.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1
echo rc=$?
tail -c 400 $O/plate_h3.log; tail -c 300 $O/plate_1m.log; tail -c 300 $O/emb3.log
