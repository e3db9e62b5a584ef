#!/bin/bash
# headline MSM (2^20 points, four in flight) under environment settings: tools/msm_sweep.sh OUT "A=1" "A=2 B=3" ...
out=$1; shift; : > $out
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python bench.py --no-proofs --no-big-proof --no-batch --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('$cfg', 'c', j['config']['window_bits'], 'ms', round(j['ms_per_step'],4), 'acc_in_flight_us', round(r['kernel_avg_us'],1), 'acc_alone_us', round(r['kernel_alone_us'],1), 'frac', round(r['frac'],4), j['config']['result'][:8])" >> $out || exit 1
done
