#!/bin/bash
# round 4: coarse parts of a pair of solves side by side (FCVM_PAIR_OVERLAP) -- bitwise test, pair time, GNLY probe, both ways
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "two_right_hand_sides or buckling_factors" > gpurun_out/r04x_tests.log 2>&1 || { tail -40 gpurun_out/r04x_tests.log; exit 1; }
tail -3 gpurun_out/r04x_tests.log
timeout -k 10 300 python tools/sessions/pair_time_r04.py > gpurun_out/r04x_pair.log 2>&1
FCVM_PAIR_OVERLAP=0 timeout -k 10 300 python tools/sessions/pair_time_r04.py > gpurun_out/r04x_pair_off.log 2>&1
tail -3 gpurun_out/r04x_pair.log; tail -3 gpurun_out/r04x_pair_off.log
timeout -k 10 300 python tools/gnly_probe.py > gpurun_out/r04x_gnly.log 2>&1
FCVM_PAIR_OVERLAP=0 timeout -k 10 300 python tools/gnly_probe.py > gpurun_out/r04x_gnly_off.log 2>&1
tail -4 gpurun_out/r04x_gnly.log; tail -4 gpurun_out/r04x_gnly_off.log
