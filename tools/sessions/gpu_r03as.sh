#!/bin/bash
# round 3: smoother parameters around the new aggregate size (whole analyses of the 1 M plate)
O=gpurun_out/r03as
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo,lo:env.FCVM_MG_ALPHA1S=8,lo:env.FCVM_MG_ALPHA1S=20,lo:env.FCVM_MG_ALPHA1S=40,lo:w=0.9,lo:w=1.1,lo,lo:alpha0=8,lo:alpha0=32,lo:env.FCVM_MG_K1S=4;env.FCVM_MG_ALPHA1S=30,lo"
timeout -k 10 700 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.txt 2>&1 || exit 1
python3 - <<'PY'
import json
for r in json.load(open('gpurun_out/r03as/plate_1m.json')):
    print("%-48s its %.1f  ms/it %.3f  solve %.2f  analysis %.2f" % (r['variant'], r.get('its_per_solve',0), r.get('ms_per_pcg_iteration',0), r.get('solve_s',0), r.get('analysis_s',0)), r.get('error','')[:80])
PY
