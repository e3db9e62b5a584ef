set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -- python3 bench.py --no-proofs --no-big-proof --no-batch --no-cpu-baseline > $R/gpurun_out/prof_cur.log 2>&1)
