"""Timeline of ONE hardware queue (one job stream) of the MSM bench from a kernel trace, with the passes of the other
queues marked: python tools/msm_queue_timeline.py TRACE_DIR"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bpg::", "").replace("void ", ""), r["Queue_Id"])
        for r in csv.DictReader(open(f))]
rows.sort()
acc = [r for r in rows if r[2].startswith("k_accumulate")]
q = acc[len(acc) // 2][3]
t0 = acc[len(acc) // 2 - 4][0]
t1 = acc[len(acc) // 2 + 5][1]
print("queue", q, "(us from the start of an earlier pass); passes of every queue:")
for a in acc:
    if t0 <= a[0] <= t1:
        print("   pass on queue %s: %8.1f .. %8.1f" % (a[3], (a[0] - t0) / 1e3, (a[1] - t0) / 1e3))
print("kernels of queue", q)
for s, e, n, qq in rows:
    if qq == q and t0 <= s <= t1:
        print("   %-24s %8.1f .. %8.1f  (%.1f us)" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
