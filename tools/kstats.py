"""Print per-kernel average durations from a rocprofv3 --kernel-trace --stats CSV directory."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("bpg::", "")
        print("%-28s calls %5s avg %9.1f us total %10.1f us %6s%%" % (name[:28], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                     float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
