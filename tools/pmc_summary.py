"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py DIR [proofs]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
proofs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[k] += 1
allv = sum(v.get("SQ_INSTS_VALU", 0) for v in tot.values())
print("%-40s %8s %14s %7s %12s" % ("kernel", "calls", "VALU/proof", "share", "waves/proof"))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    print("%-40s %8d %14.0f %6.1f%% %12.0f" % (k[:40], calls[k], v.get("SQ_INSTS_VALU", 0) / proofs, 100 * v.get("SQ_INSTS_VALU", 0) / allv,
                                            v.get("SQ_WAVES", 0) / proofs))
print("total VALU wave-instructions per proof: %.0f" % (allv / proofs))
