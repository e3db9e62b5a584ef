#!/bin/bash
# bench.py's secondary metric (and the headline) under a list of environment settings: tools/sweep_env.sh OUT "A=1 B=2" "A=2" ...
out=$1; shift; : > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python bench.py --no-big-proof --no-batch --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['secondary']
print('msm_ms', round(j['ms_per_step'],4), 'prove', round(s['prove'],1), 'p+v', round(s['prove_and_verify'],1), 'lat', s['single_proof_latency_ms']['prove'], 'cpu', round(s['host_cpu_ms_per_proof']['prove'],2))" >> $out || exit 1
done
