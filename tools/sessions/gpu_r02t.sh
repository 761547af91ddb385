#!/bin/bash
# round-2 GPU session T: window format with 16-byte component pairs and line-aligned waves
O=gpurun_out/r02t
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python3 tools/spmv_ab.py --cells 55 --rounds 8 --variants FCVM_SPMV_WIN=0 FCVM_WIN_V2=1 FCVM_WIN_V2=0 > $O/ab_win.log 2>&1 || { tail -20 $O/ab_win.log; exit 1; }
cat $O/ab_win.log | cut -c1-250
timeout -k 10 400 python3 tools/spmv_ab.py --mesh embankment3 --rounds 8 --variants FCVM_SPMV_WIN=0 FCVM_WIN_V2=1 FCVM_WIN_V2=0 > $O/ab_win_emb.log 2>&1 || { tail -20 $O/ab_win_emb.log; exit 1; }
cat $O/ab_win_emb.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "window or pcg or precond or cube or lug or embankment" > $O/parity.log 2>&1; tail -3 $O/parity.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"back_to_back_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"stream_read_GBs_this_box": [0-9.]*\|"assembly": {"ms": [0-9.]*' $O/bench.log
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
run ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
grep -A3 "k_spmv_win\|^elements\|^spmv kernel\|^stream" $O/pmc_summary.txt | cut -c1-160
