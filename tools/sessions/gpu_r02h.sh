#!/bin/bash
O=gpurun_out/r02h
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "assemble or tangent or cube_analysis or embankment_analysis or bitwise or gnly_updated or buckling_factors or lug_pressure or u100_column_full" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
FCVM_ASSEMBLE_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_seq -- python3 tools/spmv_probe.py --reps 2 --pcg 0 > $O/trace_seq.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_ovl -- python3 tools/spmv_probe.py --reps 2 --pcg 0 > $O/trace_ovl.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gnly -- python3 tools/example_run.py embankment --refine 2 --nstep 6 --gnl GNLY > $O/gnly.log 2>&1
tail -3 $O/gnly.log
echo done
