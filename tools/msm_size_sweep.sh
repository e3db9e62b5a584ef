#!/usr/bin/env bash
# MSM points/s for n = 2^16 ... 2^22 (BASELINE config 2's sweep; headline 2^20), 4 MSMs in flight, resident scalars.
# Run on the GPU box from the repo root; one line per size.
set -euo pipefail
O=gpurun_out/sweep
mkdir -p $O
for logn in 16 17 18 19 20 21 22; do
  timeout -k 10 200 python3 bench.py --no-proofs --no-big-proof --no-batch --no-extras --logn $logn --steps 30 > $O/n_$logn.json 2> $O/n_$logn.err
  python3 -c "
import json
d=json.loads(open('$O/n_$logn.json').read().strip().splitlines()[-1])
r=d['roofline']; c=d['cpu_baseline']
print('2^%d: %.1f M points/s, %.3f ms per MSM, window %d bits x %d, k_accumulate %.1f us (alone %.1f), CPU port %d cores %.2f M points/s (1 core %.3f), parity %s' % ($logn, d['value']/1e6, d['ms_per_step'], d['config']['window_bits'], d['config']['windows'], r['kernel_avg_us'], r['kernel_alone_us'], c['cores'], c['value']/1e6, c['single_core']['value']/1e6, c['parity_with_gpu']))"
done
