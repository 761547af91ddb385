#!/bin/bash
# round 4: assembly time against the number of hardware queues (seven length classes on concurrent streams)
mkdir -p gpurun_out
for q in default 8 2 1; do
  for mode in elastic tangent; do
    arg=""; [ $mode = tangent ] && arg=tangent
    if [ $q = default ]; then
      ROUTES=direct timeout -k 10 300 python3 tools/sessions/asm_time_r04.py $arg > gpurun_out/r04zj_$mode.$q.log 2>&1
    else
      GPU_MAX_HW_QUEUES=$q ROUTES=direct timeout -k 10 300 python3 tools/sessions/asm_time_r04.py $arg > gpurun_out/r04zj_$mode.$q.log 2>&1
    fi
    echo "queues $q: $(grep calc_gsm gpurun_out/r04zj_$mode.$q.log | cut -c1-100)"
  done
done
