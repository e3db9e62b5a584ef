#!/usr/bin/env bash
# headline MSM bench: hardware queues x in-flight depth, alternating to average out drift
set -euo pipefail
: > gpurun_out/hwq.log
for r in 1 2 3; do
for q in 8 16; do
for k in 4 5; do
    echo "== hwq $q in-flight $k" >> gpurun_out/hwq.log
    BPG_BENCH_HW_QUEUES=$q timeout -k 10 100 python bench.py --in-flight $k --no-proofs --no-big-proof --no-batch --no-cpu-baseline >> gpurun_out/hwq.log 2>&1
done
done
done
