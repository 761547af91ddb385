"""one tangent refresh of the GNLY probe from a rocprofv3 kernel trace (csv): every dispatch from the first assembly kernel of a
refresh to the first FP64 SpMV after it (start offset, duration, queue)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_assemble_lds<true" in r["Kernel_Name"]]
groups, last = [], None
for i in idx:
    if last is None or int(rows[i]["Start_Timestamp"]) - int(rows[last]["Start_Timestamp"]) > 20e6:
        groups.append(i)
    last = i
i0 = groups[min(int(sys.argv[2]) if len(sys.argv) > 2 else 3, len(groups) - 1)]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(i0 - 6, 0):]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %8.1f  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
    if "k_spmv_win_sym" in r["Kernel_Name"] and s > 0:
        break
