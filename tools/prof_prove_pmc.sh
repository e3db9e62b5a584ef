#!/bin/bash
# Per-kernel VALU instruction counts of proving at 2^16 multipliers (one thread, groups of eight): where the GPU's
# issue slots go.  Output: gpurun_out/r3_prove_pmc/**/ *_counter_collection.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r3_prove_pmc -- python3 tools/prove_many_timing.py 8 1 > $R/gpurun_out/r3_prove_pmc.log 2>&1)
