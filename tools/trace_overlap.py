"""Reads a rocprofv3 kernel_trace.csv and prints, for consecutive k_tower dispatches, start/end and overlap."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tw = [r for r in rows if "k_tower" in r["Kernel_Name"]]
t0 = int(tw[len(tw) // 2]["Start_Timestamp"])
for r in tw[len(tw) // 2: len(tw) // 2 + 8]:
    print("tower q%s start %8.1f us end %8.1f us dur %7.1f grid %s" % (r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
lo, hi = int(tw[len(tw) // 2]["Start_Timestamp"]), int(tw[len(tw) // 2 + 3]["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= lo and e <= hi and "k_tower" not in r["Kernel_Name"]:
        print("   %-28s q%s start %8.1f end %8.1f dur %6.1f" % (r["Kernel_Name"].split("<")[0].replace("void ", "")[:28], r["Queue_Id"], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
