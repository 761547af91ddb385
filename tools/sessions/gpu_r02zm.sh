#!/bin/bash
# round-2 GPU session ZM: HBM traffic of the final SpMV layout (FETCH_SIZE / WRITE_SIZE, separate passes, with the stream-read calibration)
O=gpurun_out/r02zm
mkdir -p $O/pmc
export TMPDIR=/tmp
PROBE="python3 tools/spmv_probe.py --reps 6 --pcg 0 --membench"
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc/$name -- $PROBE > $O/pmc_$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name hung: stopping"; exit 1; fi
}
timeout -k 10 240 python3 tools/spmv_probe.py --reps 30 --pcg 0 --membench > $O/pmc/plain.log 2>&1 || exit 1
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
grep -A2 "k_spmv_win\|k_stream_read\|^elements\|^spmv kernel\|^stream" $O/pmc_summary.txt | cut -c1-160
