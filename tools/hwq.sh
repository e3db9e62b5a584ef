#!/usr/bin/env bash
set -euo pipefail
: > gpurun_out/hwq.log
for q in 8 16; do
for k in 4 6; do
    echo "== hwq $q in-flight $k" >> gpurun_out/hwq.log
    GPU_MAX_HW_QUEUES=$q timeout -k 10 100 python bench.py --in-flight $k --no-proofs --no-big-proof --no-batch --no-cpu-baseline >> gpurun_out/hwq.log 2>&1
done
done
for q in 4 8 16; do
    echo "== proofs hwq $q" >> gpurun_out/hwq.log
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --in-flight 2 --no-big-proof --no-batch --no-cpu-baseline >> gpurun_out/hwq.log 2>&1
done
