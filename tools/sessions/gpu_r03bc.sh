#!/bin/bash
O=gpurun_out/r03bc
mkdir -p $O
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "set_up_switches" > $O/tests.log 2>&1; echo tests rc=$?; tail -8 $O/tests.log
