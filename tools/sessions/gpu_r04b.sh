#!/bin/bash
# round 4, second GPU call: two-vector kernels on the symmetric window form (bitwise test, pair timing), the dist test that failed in r04a
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "two_right_hand_sides or reruns_are_bitwise or window_format or pcg_solve" > $O/tests_pair.log 2>&1; echo "rc=$?" >> $O/tests_pair.log; tail -3 $O/tests_pair.log
python -m pytest tests/test_gpu_dist.py -x -q > $O/tests_dist.log 2>&1; echo "rc=$?" >> $O/tests_dist.log; tail -3 $O/tests_dist.log
python tools/sessions/pair_time_r04.py > $O/pair_time.log 2>&1; tail -4 $O/pair_time.log
FCVM_SYM_PAIR=0 python tools/sessions/pair_time_r04.py > $O/pair_time_off.log 2>&1; tail -2 $O/pair_time_off.log
python tools/gnly_probe.py --cells 55 > $O/gnly_probe.log 2>&1; tail -7 $O/gnly_probe.log
