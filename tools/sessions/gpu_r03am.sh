#!/bin/bash
O=gpurun_out/r03am
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/prof -o g -- python3 $R/tools/gnly_probe.py --steps 4 > $R/$O/gnly.log 2>&1; echo rc=$?
cd $R
python3 tools/rocpd_timeline.py $O/prof/g_results.db k_assemble_lds 3 > $O/timeline.txt 2> $O/timeline.err; head -3 $O/timeline.err; head -48 $O/timeline.txt | cut -c1-130
rm -f $O/prof/*.db
