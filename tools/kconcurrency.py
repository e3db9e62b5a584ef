"""GPU occupancy over time from a rocprofv3 kernel trace taken under load (many prover threads): how much of the wall
time has an accumulate pass in flight (the only kernel that fills the chip), how much has only small kernels, how
much has nothing.  usage: kconcurrency.py <trace dir> [skip_fraction]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
t_lo, t_hi = min(r[0] for r in rows), max(r[1] for r in rows)
t0 = t_lo + skip * (t_hi - t_lo)                 # skip set-up and the first (allocating) pass
ev = []
for s, e, name in rows:
    if e <= t0:
        continue
    s = max(s, t0)
    big = "k_accumulate" in name
    ev.append((s, 1, big))
    ev.append((e, -1, big))
ev.sort()
n_all = n_big = 0
last = t0
t_none = t_small = t_big = 0
conc_area = 0
for t, d, big in ev:
    dt = t - last
    if n_big:
        t_big += dt
    elif n_all:
        t_small += dt
    else:
        t_none += dt
    conc_area += dt * n_all
    last = t
    n_all += d
    if big:
        n_big += d
tot = t_none + t_small + t_big
print("window %.1f ms: accumulate in flight %.1f%%, only other kernels %.1f%%, nothing %.1f%%, mean kernels in flight %.2f"
      % (tot / 1e6, 100 * t_big / tot, 100 * t_small / tot, 100 * t_none / tot, conc_area / tot))
