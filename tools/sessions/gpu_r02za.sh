#!/bin/bash
# round-2 GPU session ZA: sparse aggregate level with 8 fused smoothing steps: forced test, 1 M cube probe, 8 M cube
O=gpurun_out/r02za
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "sparse_aggregate or pcg or precond" > $O/parity.log 2>&1; tail -5 $O/parity.log | cut -c1-220
timeout -k 10 300 python3 tools/l2s_probe.py --cube 55 --max-aggr 200 > $O/probe_cube.log 2>&1; grep -v "window format\|x.Mx" $O/probe_cube.log | head -8 | cut -c1-200
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
for sp in 1 0; do
  FCVM_MG_SPARSE=$sp timeout -k 10 420 python3 bench.py --cells 110 --steps 4 --warmup 2 --no-cpu --no-plate > $O/bench8m_sparse$sp.log 2> $O/bench8m_sparse$sp.err || { tail -5 $O/bench8m_sparse$sp.err; exit 1; }
  echo "sparse $sp"; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"assembly_plus_preconditioner_hierarchy": [0-9.]*\|aggregates x[^"]*' $O/bench8m_sparse$sp.log
done
