#!/bin/bash
# round 3: aggregate size x level-1 steps (whole analyses of the 1 M plate)
O=gpurun_out/r03ao
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo:env.FCVM_MG_TARGET=72,lo:env.FCVM_MG_TARGET=72;env.FCVM_MG_K1S=4,lo:env.FCVM_MG_TARGET=64;env.FCVM_MG_K1S=4,lo:env.FCVM_MG_TARGET=64;env.FCVM_MG_K1S=2,lo:env.FCVM_MG_TARGET=56,lo:env.FCVM_MG_TARGET=48,lo:env.FCVM_MG_TARGET=64;env.FCVM_MG_K1S=4;env.FCVM_MG_ALPHA1S=30"
timeout -k 10 600 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.txt 2>&1 || exit 1
python3 - <<'PY'
import json
for r in json.load(open('gpurun_out/r03ao/plate_1m.json')):
    print("%-70s its %.1f  ms/it %.3f  solve %.2f  analysis %.2f  %s" % (r['variant'], r.get('its_per_solve',0), r.get('ms_per_pcg_iteration',0), r.get('solve_s',0), r.get('analysis_s',0), r.get('precond', r.get('error',''))[:60]))
PY
