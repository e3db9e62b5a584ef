#!/bin/bash
# proofs/s against the number of prover threads (groups of eight): tools/thread_sweep.sh OUT T1 T2 ...
out=$1; shift; : > $out
for t in "$@"; do
  timeout -k 10 170 python bench.py --no-big-proof --no-batch --no-cpu-baseline --no-extras --steps 5 --proof-threads $t 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['secondary']; print('threads $t', round(s['prove'],1), round(s['prove_and_verify'],1), 'cpu', round(s['host_cpu_ms_per_proof']['prove'],2))" >> $out || exit 1
done
