#!/bin/bash
# round 3: aggregate size once more with the symmetric product (whole analyses of the 1 M plate, one process per variant list)
O=gpurun_out/r03ar
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo,lo:env.FCVM_MG_TARGET=64,lo:env.FCVM_MG_TARGET=80,lo:env.FCVM_L2_SYMV=0,lo:env.FCVM_MG_K1S=4,lo"
timeout -k 10 600 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.txt 2>&1 || exit 1
python3 - <<'PY'
import json
for r in json.load(open('gpurun_out/r03ar/plate_1m.json')):
    print("%-40s its %.1f  ms/it %.3f  solve %.2f  analysis %.2f  %s" % (r['variant'], r.get('its_per_solve',0), r.get('ms_per_pcg_iteration',0), r.get('solve_s',0), r.get('analysis_s',0), r.get('precond', r.get('error',''))[:60]))
PY
