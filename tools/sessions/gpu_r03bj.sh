#!/bin/bash
# round 3: symmetric repack kernel with prefetched source block numbers -- parity + kernel time
O=gpurun_out/r03bj
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "symmetric_window or window_format_spmv or reruns" > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 10 --solves 1 > $R/$O/lo.txt 2>&1 || exit 1
python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 40 > $R/$O/lo_kernels.txt
rm -f $R/$O/prof/*.db
grep "repack\|assemble_lds\|absmax\|diag_get\|diag_invert\|dinv_pack" $R/$O/lo_kernels.txt | cut -c1-50,112-170
