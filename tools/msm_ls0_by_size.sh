#!/bin/bash
# MSM throughput by size and entries per accumulate thread (four in flight): tools/msm_ls0_by_size.sh OUT
out=$1; : > $out
for logn in 16 17 18 19 20 21 22; do
  for l in 16 32 48; do
    BPG_MSM_LS0=$l timeout -k 10 200 python bench.py --no-proofs --no-big-proof --no-batch --no-cpu-baseline --no-extras --logn $logn --steps 30 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('2^$logn S0=$l c', j['config']['window_bits'], 'ms', round(j['ms_per_step'],4), 'Mpts/s', round(j['value']/1e6,1), 'acc_in_flight_us', round(r['kernel_avg_us'],1), 'alone', round(r['kernel_alone_us'],1))" >> $out || exit 1
  done
done
