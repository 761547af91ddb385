#!/bin/bash
# round 3: GNLY iteration cost and the Embankment analysis at the final defaults
O=gpurun_out/r03ax
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/gnly_probe.py --cells 55 > $O/gnly.txt 2>&1; echo gnly rc=$?; tail -5 $O/gnly.txt
timeout -k 10 300 python3 tools/example_run.py embankment --refine 3 > $O/emb3.txt 2>&1; echo emb rc=$?; tail -6 $O/emb3.txt | cut -c1-300
