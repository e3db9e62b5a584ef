#!/bin/bash
set -e
T=${1:-24}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && GPU_MAX_HW_QUEUES=8 timeout -k 10 250 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $R/gpurun_out/r3_lat_t$T -- python3 tools/prove_many_timing.py 8 $T > $R/gpurun_out/r3_lat_t$T.log 2>&1)
cd $R && python3 tools/klatency.py gpurun_out/r3_lat_t$T > gpurun_out/r3_lat_t$T.txt
rm -rf gpurun_out/r3_lat_t$T
