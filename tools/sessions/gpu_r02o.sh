#!/bin/bash
# round-2 GPU session O: clock / power log of the SpMV (back to back and inside the solver), PMC of the shipped layout
# (FETCH / WRITE for profiles/spmv_traffic.json, L1 -> L2 request counts and L2 hit rates)
O=gpurun_out/r02o
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/clock_log.py --what spmv --seconds 15 > $O/clock_spmv.log 2>&1 || { tail $O/clock_spmv.log; exit 1; }
head -4 $O/clock_spmv.log | cut -c1-300; tail -30 $O/clock_spmv.log
timeout -k 10 200 python3 tools/clock_log.py --what pcg --seconds 15 > $O/clock_pcg.log 2>&1 || { tail $O/clock_pcg.log; exit 1; }
tail -30 $O/clock_pcg.log
PROBE="python3 tools/spmv_probe.py --reps 6 --pcg 0 --membench"
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc/$name -- $PROBE > $O/pmc_$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name hung: stopping"; exit 1; fi
}
mkdir -p $O/pmc
timeout -k 10 240 python3 tools/spmv_probe.py --reps 30 --pcg 0 --membench > $O/pmc/plain.log 2>&1 || exit 1
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run tcp3 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
