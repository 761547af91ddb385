#!/bin/bash
# round-2 GPU session X: 8 M-element cube: how many aggregates should the dense third level have? (default: 1000)
O=gpurun_out/r02x
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
for tg in default 457 300; do
  if [ $tg = default ]; then unset FCVM_MG_TARGET; else export FCVM_MG_TARGET=$tg; fi
  SECONDS=0; timeout -k 10 420 python3 bench.py --cells 110 --steps 4 --warmup 2 --no-cpu --no-plate > $O/bench8m_$tg.log 2> $O/bench8m_$tg.err || { tail -5 $O/bench8m_$tg.err; exit 1; }
  echo "target $tg"; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"assembly_plus_preconditioner_hierarchy": [0-9.]*\|aggregates x[^"]*' $O/bench8m_$tg.log; echo "wall $SECONDS s"
done
