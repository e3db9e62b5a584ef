"""Timeline of the k-th MSM (k-th k_digits launch .. the next one) from a rocprofv3 kernel trace CSV."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_digits" in r["Kernel_Name"]]
s, e = idx[k], idx[k + 1]
t0 = int(rows[s]["Start_Timestamp"])
tot = 0
for r in rows[s:e]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%-26s start %8.1f dur %8.1f us grid %8s" % (r["Kernel_Name"].split("(")[0].replace("bpg::", "").replace("void ", "")[:26],
                                                      (int(r["Start_Timestamp"]) - t0) / 1e3, d, r["Grid_Size_X"]))
print("sum of kernel durations: %.1f us" % tot)
