"""Launch-to-start delay and duration per kernel from a rocprofv3 run with --kernel-trace --hip-trace (many prover
threads): is a round's latency queueing in front of the GPU, or execution on it?
usage: klatency.py <trace dir>"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
at = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
api = {}
for r in csv.DictReader(open(at)):
    if "Launch" in r["Function"]:
        api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = list(csv.DictReader(open(kt)))
t_lo = min(int(r["Start_Timestamp"]) for r in rows)
t_hi = max(int(r["End_Timestamp"]) for r in rows)
t0 = t_lo + 0.4 * (t_hi - t_lo)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or r["Correlation_Id"] not in api:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("bpg::", "").replace("void ", "")[:28]
    a = agg[name]
    a[0] += 1
    a[1] += (s - api[r["Correlation_Id"]][0]) / 1e3
    a[2] += (e - s) / 1e3
    a[3] = max(a[3], (s - api[r["Correlation_Id"]][0]) / 1e3)
print("%-28s %7s %12s %12s %12s" % ("kernel", "calls", "delay_us", "dur_us", "max_delay"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %7d %12.1f %12.1f %12.1f" % (k, a[0], a[1] / a[0], a[2] / a[0], a[3]))

# per hardware queue: how busy, and do its kernels ever overlap (a queue that runs one kernel at a time is a FIFO
# server: everything submitted to it waits for everything before it, whatever stream it came from)
qs = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0:
        qs[r["Queue_Id"]].append((s, e))
span = t_hi - t0
print("\n%-8s %8s %10s %12s %14s" % ("queue", "kernels", "busy_frac", "overlap_frac", "mean_gap_us"))
for q, iv in sorted(qs.items()):
    iv.sort()
    busy = ov = 0
    last_e = iv[0][0]
    gaps = []
    for s, e in iv:
        if s < last_e:
            ov += min(e, last_e) - s
            busy += max(0, e - last_e)
        else:
            gaps.append(s - last_e)
            busy += e - s
        last_e = max(last_e, e)
    print("%-8s %8d %10.3f %12.3f %14.1f" % (q, len(iv), busy / span, ov / span, (sum(gaps) / len(gaps) / 1e3) if gaps else 0))
