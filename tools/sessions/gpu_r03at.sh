#!/bin/bash
# round 3: diagonal-tile Cholesky kernel rewritten -- LAPACK parity, set-up stage times at the new aggregate size
O=gpurun_out/r03at
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests -q -m gpu -x -k "dense_spd or lower_triangle or sparse_aggregate or preconditioners" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
FCVM_SETUP_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup.txt 2>&1; grep "fcvm set-up\|RiksDriver" $O/setup.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o s -- python3 $R/tools/setup_inside.py > $R/$O/setup_prof.txt 2>&1
python3 $R/tools/rocpd_top.py $R/$O/prof/s_results.db 40 > $R/$O/setup_kernels.txt; rm -f $R/$O/prof/*.db
grep "chol\|trinv\|xtx\|galerkin\|sym_pad\|copy_block" $R/$O/setup_kernels.txt | cut -c1-50,112-170
