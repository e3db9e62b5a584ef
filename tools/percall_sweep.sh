#!/bin/bash
# proofs/s against the statements per prove_many call: tools/percall_sweep.sh OUT K1 K2 ...
out=$1; shift; : > $out
for k in "$@"; do
  timeout -k 10 170 python bench.py --no-big-proof --no-batch --no-cpu-baseline --no-extras --steps 5 --proofs-per-call $k 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['secondary']; print('per call $k', round(s['prove'],1), round(s['prove_and_verify'],1), 'cpu', round(s['host_cpu_ms_per_proof']['prove'],2))" >> $out || exit 1
done
