set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
lib=$R/bulletproofs_gadgets_amd/libbpg.so
cp $lib $R/ab/libbpg_new.so
for which in old new; do
  cp $R/ab/libbpg_$which.so $lib
  (cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$which -- python3 bench.py --no-proofs --no-big-proof --no-batch --no-cpu-baseline > $R/gpurun_out/prof_$which.log 2>&1)
done
cp $R/ab/libbpg_new.so $lib
