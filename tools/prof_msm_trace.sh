#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/msm_trace -- python3 bench.py --no-proofs --no-big-proof --no-batch --no-cpu-baseline --no-extras --steps 30 > $R/gpurun_out/msm_trace.log 2>&1)
cd $R && python3 tools/msm_gaps.py gpurun_out/msm_trace > gpurun_out/msm_gaps.txt
python3 tools/msm_queue_timeline.py gpurun_out/msm_trace > gpurun_out/msm_queue_timeline.txt
rm -f gpurun_out/msm_trace/*/*kernel_trace.csv
