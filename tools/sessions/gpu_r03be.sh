#!/bin/bash
# round 3: kernel table of the Embankment 337k analysis (where does a 0.5 ms PCG iteration go on a mid-size mesh?)
O=gpurun_out/r03be
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o e -- python3 $R/tools/example_run.py embankment --refine 3 > $R/$O/emb.txt 2>&1; echo rc=$?
python3 $R/tools/rocpd_top.py $R/$O/prof/e_results.db 26 > $R/$O/emb_kernels.txt; rm -f $R/$O/prof/*.db
tail -1 $R/$O/emb.txt | cut -c1-330; cut -c1-56,112-170 $R/$O/emb_kernels.txt
