#!/bin/bash
# K processes x T prover threads on ONE GPU: is a process (its HIP runtime, its interpreter) what limits proofs/s?
# usage: tools/multiproc_proofs.sh K T [extra env assignments...]; prints each process's proofs/s and their sum
K=$1; T=$2; shift 2
out=gpurun_out/mp_${K}x${T}
mkdir -p $out; rm -f $out/*.json
pids=()
for i in $(seq 1 $K); do
  env "$@" BPG_BENCH_PROOF_THREADS=$T timeout -k 10 300 python bench.py --no-big-proof --no-batch --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $out/p$i.json 2> $out/p$i.err &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
python - $out <<'PY'
import glob, json, sys
tot = 0; tv = 0
for f in sorted(glob.glob(sys.argv[1] + "/p*.json")):
    try:
        s = json.loads(open(f).read())["secondary"]
        print(f, round(s["prove"], 1), round(s["prove_and_verify"], 1), "cpu", round(s["host_cpu_ms_per_proof"]["prove"], 2))
        tot += s["prove"]; tv += s["prove_and_verify"]
    except Exception as e:
        print(f, "ERR", e)
print("sum prove %.1f  prove+verify %.1f" % (tot, tv))
PY
exit $rc
