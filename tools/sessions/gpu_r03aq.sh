#!/bin/bash
O=gpurun_out/r03aq
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests -q -m gpu -x -k "lower_triangle" > $O/tests.log 2>&1; echo tests rc=$?; tail -12 $O/tests.log
