#!/bin/bash
# K native prover processes x T threads on one GPU at once: tools/native_multiproc.sh K T [R]
K=$1; T=$2; R=${3:-8}
python tools/dump_statement.py 2427 /tmp/stmt.bin || exit 1
pids=()
for i in $(seq 1 $K); do
  GPU_MAX_HW_QUEUES=${Q:-8} timeout -k 10 200 tools/native/prove_driver /tmp/stmt.bin $T $R 8 > /tmp/nd_$i.json 2>/tmp/nd_$i.err &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
cat /tmp/nd_*.json | python -c "
import sys, json
rs=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$K x $T:', [r['proofs_per_s'] for r in rs], 'sum', round(sum(r['proofs_per_s'] for r in rs),1), 'cpu', [r['host_cpu_ms_per_proof'] for r in rs])"
rm -f /tmp/nd_*.json /tmp/nd_*.err
exit $rc
