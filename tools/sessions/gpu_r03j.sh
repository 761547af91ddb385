#!/bin/bash
O=gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in "" _lb6 _if4; do
  FCVM_HIP_LIB=$R/fcvm-workbench_amd/fcvm_hip/libfcvm_hip$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof$v -o lo -- python3 $R/tools/lo_probe.py --solves 2 > $R/$O/lo_probe$v.log 2>&1 || exit 1
  echo "== variant '$v'"; grep "cycle form\|solve:" $R/$O/lo_probe$v.log
  python3 $R/tools/rocpd_top.py $R/$O/prof$v/lo_results.db 3 | cut -c1-60,105-160
  rm -f $R/$O/prof$v/*.db
done
cd $R
FCVM_PATTERN_TIMING=1 FCVM_VERBOSE=1 timeout -k 10 200 python3 tools/setup_profile.py --plate-h 1.45 > $O/setup_profile.log 2>&1; echo rc=$?
grep -v "^ \|^$" $O/setup_profile.log | head -40
timeout -k 10 300 python3 bench.py --plate-h 8 --cpu-plate-h 10 --cells 8 --steps 3 --warmup 2 > $O/bench_small.json 2> $O/bench_small.err; echo bench small rc=$?; tail -c 300 $O/bench_small.err; head -c 600 $O/bench_small.json
timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "16bit_smoother" 2>&1 | tail -3
