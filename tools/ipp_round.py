"""Kernel timeline of one inner-product round (k_ipp_rows .. next k_ipp_rows) from a rocprofv3 kernel trace."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_ipp_rows" in r["Kernel_Name"]]
s, e = idx[-10], idx[-9]
t0 = int(rows[s]["Start_Timestamp"])
tot = 0
for r in rows[s:e + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%-26s start %8.1f dur %7.1f grid %8s" % (r["Kernel_Name"].split("(")[0].replace("bpg::", "").replace("void ", "")[:26],
                                                   (int(r["Start_Timestamp"]) - t0) / 1e3, d, r["Grid_Size_X"]))
