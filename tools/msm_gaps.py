"""The MSM bench's pipeline (four in flight) from a rocprofv3 kernel trace: per hardware queue (= one job's stream when
streams do not share queues) the chain digits .. segmap -> accumulate -> reduce .. tail; for every pass, how long its own
sort had been finished when it started, and what ran between the previous pass's end and its start.
    python tools/msm_gaps.py TRACE_DIR"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rd = list(csv.DictReader(open(f)))
qkey = "Queue_Id" if "Queue_Id" in rd[0] else "Stream_Id"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bpg::", "").replace("void ", ""), r[qkey])
        for r in rd]
rows.sort()
acc = [r for r in rows if r[2].startswith("k_accumulate")]
acc = acc[len(acc) // 3:]
print("queues used:", sorted({r[3] for r in rows}), "(column %s)" % qkey)
own_wait, gaps, digits_lead = [], [], []
inside = defaultdict(float)
for prev, cur in zip(acc, acc[1:]):
    gaps.append(max(0.0, (cur[0] - prev[1]) / 1e3))
    mine = [r for r in rows if r[3] == cur[3] and r[1] <= cur[0]]
    seg = [r for r in mine if r[2].startswith("k_segmap")]
    dig = [r for r in mine if r[2].startswith("k_digits")]
    if seg:
        own_wait.append((cur[0] - seg[-1][1]) / 1e3)
    if dig:
        digits_lead.append((cur[0] - dig[-1][0]) / 1e3)
    for s, e, name, q in rows:
        if e <= prev[1] or s >= cur[0] or name.startswith("k_accumulate"):
            continue
        inside[name + (" [next pass's job]" if q == cur[3] else "")] += (min(e, cur[0]) - max(s, prev[1])) / 1e3
n = len(gaps)
period = (acc[-1][0] - acc[0][0]) / 1e3 / (len(acc) - 1)
print("passes %d, period %.1f us, pass %.1f us, gap mean %.1f us (min %.1f, max %.1f)" % (
    len(acc), period, sum((a[1] - a[0]) for a in acc) / 1e3 / len(acc), sum(gaps) / n, min(gaps), max(gaps)))
print("a pass starts %.1f us (mean; min %.1f, max %.1f) after its own job's k_segmap ended, %.1f us after its k_digits started" % (
    sum(own_wait) / len(own_wait), min(own_wait), max(own_wait), sum(digits_lead) / len(digits_lead)))
print("kernel time inside the gaps, per gap:")
for k, v in sorted(inside.items(), key=lambda kv: -kv[1]):
    print("  %-44s %8.1f us" % (k, v / n))
