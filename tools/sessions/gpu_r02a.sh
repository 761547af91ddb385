#!/bin/bash
# round-2 GPU session A: SpMV storage-order A/B (gangs of interleaved chunks), bench lines, configs[3]/[4] tests
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
python3 tools/spmv_ab.py --cells 55 --rounds 10 --variants FCVM_SPMV_GANG=1 FCVM_SPMV_GANG=8 FCVM_SPMV_GANG=64 "FCVM_SPMV_GANG=1,FCVM_VALS_TRIES=1" "FCVM_SPMV_GANG=64,FCVM_VALS_TRIES=1" > $O/ab_cube.log 2>&1 &&
python3 tools/spmv_ab.py --mesh embankment3 --rounds 10 --variants FCVM_SPMV_GANG=1 FCVM_SPMV_GANG=64 > $O/ab_emb.log 2>&1 &&
python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_g64.log 2>&1 &&
FCVM_SPMV_GANG=1 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_g1.log 2>&1 &&
python3 -m pytest tests/test_gpu_configs.py -x -q -s > $O/configs.log 2>&1
rc=$?
tail -n 8 $O/ab_cube.log $O/ab_emb.log; tail -c 1500 $O/bench_g64.log; tail -n 30 $O/configs.log
exit $rc
