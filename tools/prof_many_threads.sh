#!/bin/bash
# kernel trace of proving at 2^16 with many host threads (default 24); then tools/kconcurrency.py on it
set -e
T=${1:-24}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && GPU_MAX_HW_QUEUES=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_trace_t$T -- python3 tools/prove_many_timing.py 8 $T > $R/gpurun_out/r3_trace_t$T.log 2>&1)
cd $R && python3 tools/kconcurrency.py gpurun_out/r3_trace_t$T > gpurun_out/r3_conc_t$T.txt
tail -3 gpurun_out/r3_trace_t$T.log | cut -c1-400 >> gpurun_out/r3_conc_t$T.txt
rm -f gpurun_out/r3_trace_t$T/*/*kernel_trace.csv    # tens of MB: only the stats travel back
