#!/bin/bash
O=gpurun_out/r03aa
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo:env.FCVM_MG_L1_16=0,lo,lo:env.FCVM_MG_K1S=4,lo:env.FCVM_MG_K1S=4;env.FCVM_MG_L1_16=0"
timeout -k 10 600 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.log 2>&1 &&
timeout -k 10 600 python3 tools/precond_probe.py --case embankment --refine 3 --variants "$V" --out $O/emb3.json > $O/emb3.log 2>&1 &&
timeout -k 10 300 python3 tools/precond_probe.py --case plate --h 4.0 --variants "$V" --out $O/plate_48k.json > $O/plate_48k.log 2>&1
echo rc=$?
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --solves 2 > $R/$O/lo_probe.log 2>&1
cd $R
grep "cycle form\|solve:" $O/lo_probe.log
python3 tools/rocpd_top.py $O/prof/lo_results.db 8 | cut -c1-60,105-160
rm -f $O/prof/*.db
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "reruns or 16bit or two_right or pcg_solve or embankment or u100" > $O/tests.log 2>&1; echo rc=$?; tail -3 $O/tests.log
