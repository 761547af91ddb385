"""one PCG iteration of a pair of solves from a rocprofv3 kernel trace (csv): start offset, duration, queue of every dispatch
between two launches of the two-vector FP64 SpMV"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_spmv_win_sym<true" in r["Kernel_Name"] and ", 2>" in r["Kernel_Name"]]
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 60
i0, i1 = idx[occ], idx[occ + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %7.1f  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
