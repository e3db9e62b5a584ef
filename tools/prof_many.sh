set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_many -- python3 tools/prove_many_timing.py 8 1 > $R/gpurun_out/prof_many.log 2>&1)
