#!/bin/bash
O=gpurun_out/r02j
mkdir -p $O
for k in 1 2 3; do
  FCVM_MG_K1S=$k timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_k1s$k.log 2>&1 || exit 1
  echo "k1s=$k"; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*' $O/bench_k1s$k.log | tr '\n' ' '; echo
done
FCVM_MG_K1S=1 timeout -k 10 300 python3 tools/example_run.py plate --h 1.45 > $O/plate_k1s1.log 2>&1; tail -1 $O/plate_k1s1.log | cut -c1-400
timeout -k 10 300 python3 tools/example_run.py plate --h 1.45 > $O/plate_k1s2.log 2>&1; tail -1 $O/plate_k1s2.log | cut -c1-400
echo done
