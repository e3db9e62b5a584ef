"""Copy the summaries of tools/refresh_profiles.sh (gpurun_out/final) into profiles/ under this round's names.
    python tools/collect_profiles.py r03
PMC files are trimmed to Kernel_Name, Counter_Name, Counter_Value, Dispatch_Id (the other columns repeat per row)."""
import csv
import glob
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
tag = sys.argv[1]


def one(pattern):
    m = glob.glob(os.path.join(F, pattern))
    return m[0] if m else None


def trim(src, dst):
    with open(src) as f, open(dst, "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id"])
        for r in csv.DictReader(f):
            w.writerow([r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"], r["Dispatch_Id"]])


copies = {"bench_n1.json": "%s_bench_n1.json", "msm_stats/*/*kernel_stats.csv": "%s_bench_msm_kernel_stats.csv",
          "prove_stats/*/*kernel_stats.csv": "%s_prove_2_16_kernel_stats.csv",
          "prove_stats_t24/*/*kernel_stats.csv": "%s_prove_2_16_24threads_kernel_stats.csv",
          "prove_concurrency_t24.txt": "%s_prove_2_16_24threads_concurrency.txt"}
for src, dst in copies.items():
    s = one(src)
    if s:
        shutil.copy(s, os.path.join(P, dst % tag))
for src, dst in {"pmc_fetch/*/*counter_collection.csv": "%s_pmc_fetch_size.csv", "pmc_write/*/*counter_collection.csv": "%s_pmc_write_size.csv",
                 "prove_pmc/*/*counter_collection.csv": "%s_prove_2_16_pmc_valu.csv"}.items():
    s = one(src)
    if s:
        trim(s, os.path.join(P, dst % tag))
if one("prove_pmc/*/*counter_collection.csv"):
    out = subprocess.run([sys.executable, os.path.join(R, "tools", "pmc_summary.py"), os.path.join(F, "prove_pmc"), "24"],
                         capture_output=True, text=True).stdout
    open(os.path.join(P, "%s_prove_2_16_valu_per_kernel.txt" % tag), "w").write(
        "VALU wave-instructions per proof, per kernel: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES over\n"
        "tools/prove_many_timing.py 8 1 (one host thread, 3 groups of eight 2^16-multiplier proofs = 24 proofs; the generator\n"
        "table's one-time kernels k_shift_table (1 of its calls), k_from_uniform, k_build_comb are in the sum, spread over 24)\n\n" + out)
for ln in open(os.path.join(F, "prove_stats.log")):
    if ln.startswith("{"):
        open(os.path.join(P, "%s_prove_2_16_profiled_run.json" % tag), "w").write(ln)
print(sorted(f for f in os.listdir(P) if f.startswith(tag)))
