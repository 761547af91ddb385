#!/bin/bash
# round 4: rehearsal of the N > 1 bench (ranks sharing GPU 0 over gloo) after this round's changes to the multi-GPU layer
set -o pipefail
O=gpurun_out/r04n; mkdir -p $O
for n in 2 3; do
  FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus $n --plate-h 5 --steps 6 --warmup 3 > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err; echo dist bench $n rc=$?
  tail -c 1500 $O/bench_${n}ranks.json; echo; tail -3 $O/bench_${n}ranks.err
done
