#!/bin/bash
# round-2 GPU session M: does the address of x / y change the SpMV time; nontemporal y stores; durations of the GPU suite
O=gpurun_out/r02m
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/vec_place_probe.py --cells 55 --pairs 6 > $O/vec_place.log 2>&1 || { tail $O/vec_place.log; exit 1; }
cat $O/vec_place.log | cut -c1-250
timeout -k 10 300 python3 tools/spmv_ab.py --cells 55 --rounds 10 --variants FCVM_SPMV_VARIANT=5 FCVM_SPMV_VARIANT=6 FCVM_SPMV_VARIANT=5 FCVM_SPMV_VARIANT=6 > $O/ab_nty.log 2>&1 || { tail $O/ab_nty.log; exit 1; }
cat $O/ab_nty.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_dist.py -x -q --durations=0 -k "buckling or two_ranks" > $O/dist.log 2>&1; tail -15 $O/dist.log
echo done
