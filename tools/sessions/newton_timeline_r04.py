"""one Newton iteration of the bench from a rocprofv3 kernel trace (csv): everything between two k_update_stress launches that is NOT one
of the PCG iteration's kernels, with start offsets and durations; the PCG kernels only as totals"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
PCG = ("k_spmv_win_sym<true", "k_spmv_win_lo_sym", "k_spmv_l1_i16", "k_symv", "k_pcg_update", "k_prolong_add", "k_restrict_init", "k_l2_restrict",
       "k_l2_prolong", "k_pcg_scalars")
idx = [i for i, r in enumerate(rows) if "k_update_stress" in r["Kernel_Name"]]
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 30
i0, i1 = idx[occ], idx[occ + 1]
t0 = int(rows[i0]["Start_Timestamp"])
tot = defaultdict(lambda: [0, 0.0])
other = 0.0
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"]
    if any(p in name for p in PCG):
        k = name.split("(")[0][:60]
        tot[k][0] += 1; tot[k][1] += (e - s) / 1e3
        continue
    other += (e - s) / 1e3
    print("%9.1f %8.1f  %s" % (s / 1e3, (e - s) / 1e3, name[:100]))
print("wall between the two stress updates: %.1f us; kernels outside the PCG iterations: %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, other))
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  PCG  %4d x %-60s %9.1f us" % (n, k, t))
