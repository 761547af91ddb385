#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc CSVs written by tools/pmc_probe.sh (one directory per counter group)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
short = lambda k: re.sub(r"\(.*", "", k.replace("void ", "").replace("fcvm::", ""))[:70]  # noqa: E731
print("rocprofv3 --pmc (one group per run, no trace domains) -- python3 tools/spmv_probe.py; averages per dispatch")
print(open(os.path.join(root, "plain.log")).read().strip())
for grp in sorted(os.listdir(root)):
    files = glob.glob(os.path.join(root, grp, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name") or row.get("Kernel_name") or ""
            grid = int(row.get("Grid_Size", "0") or 0)
            acc[(short(k), grid)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("\n== %s ==" % grp)
    for (k, grid), cs in sorted(acc.items(), key=lambda kv: -kv[0][1]):
        if not any(s in k for s in ("k_spmv", "k_stream_read", "k_assemble", "k_update_stress", "k_gather_qin", "k_repack")):
            continue
        if "k_spmv" in k and grid < 500000 and "k_spmv_l1" not in k:      # (the level-1 16-bit pass: 2900 one-wave workgroups)
            continue
        n = len(next(iter(cs.values())))
        print("%-72s grid %9d  n=%d" % (k, grid, n))
        for cn, v in sorted(cs.items()):
            print("    %-40s %18.1f" % (cn, sum(v) / len(v)))
