#!/bin/bash
# PMC passes over the SpMV / stream-read / assembly kernels: one small counter group per run (rocprofv3 --pmc only,
# no trace domains; a group that exceeds the hardware's counter slots is skipped, a hung run ends the script).
# usage: tools/pmc_probe.sh <outdir> [probe args]
O=$1; shift
mkdir -p $O
export TMPDIR=/tmp
PROBE="python3 tools/spmv_probe.py --reps 6 --pcg 0 --membench $*"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- $PROBE > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" >> $O/passes.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name hung: stopping" >> $O/passes.txt; python3 tools/pmc_summary.py $O > $O/summary.txt 2>&1; exit 1; fi
}
timeout -k 10 240 python3 tools/spmv_probe.py --reps 30 --pcg 0 --membench $* > $O/plain.log 2>&1 || exit 1
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
run ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run tcp3 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
run sq SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
