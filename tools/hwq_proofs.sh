#!/bin/bash
# proofs/s at 2^16 against the number of hardware queues (streams of different threads share them), with GPU-busy samples
cd $GRAFT_REPO_ROOT
for q in 4 8 16 24; do
  (for i in $(seq 1 12); do rocm-smi --showuse 2>/dev/null | grep -i "GPU use" ; sleep 1; done > gpurun_out/r3_hwq_use_$q.log) &
  BPG_BENCH_HW_QUEUES=$q timeout -k 10 200 python bench.py --no-big-proof --no-batch --no-cpu-baseline --no-extras --steps 10 --proof-threads 32 > gpurun_out/r3_hwq_$q.json 2> gpurun_out/r3_hwq_$q.err || exit 1
  wait
done
