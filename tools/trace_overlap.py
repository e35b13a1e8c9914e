"""Prints the timeline (µs, relative) of the kernels of the last bench step from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "dtw" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(sys.argv[2]):]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-28s q=%s start %8.1f end %8.1f" % (r["Kernel_Name"].split("(")[0][-28:], r.get("Queue_Id", "?"),
                                              (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
