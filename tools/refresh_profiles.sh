# Round-end measurement set (run on the GPU box from the repo root): full bench line, kernel stats of the MSM
# part, the two PMC passes for HBM traffic (separate runs), kernel stats of the proof path (one host thread).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
echo bench done >> $O/progress.log
cd /tmp && export TMPDIR=/tmp
MSM="--no-proofs --no-big-proof --no-batch --no-cpu-baseline --no-extras"
(cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msm_stats -- python3 bench.py $MSM > $O/msm_stats.log 2>&1)
echo stats done >> $O/progress.log
(cd $R && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $MSM --steps 6 --warmup 2 > $O/pmc_fetch.log 2>&1)
echo fetch done >> $O/progress.log
(cd $R && timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $MSM --steps 6 --warmup 2 > $O/pmc_write.log 2>&1)
echo write done >> $O/progress.log
(cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prove_stats -- python3 tools/prove_many_timing.py 8 1 > $O/prove_stats.log 2>&1)
echo prove done >> $O/progress.log
(cd $R && timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/prove_pmc -- python3 tools/prove_many_timing.py 8 1 > $O/prove_pmc.log 2>&1)
echo prove pmc done >> $O/progress.log
(cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prove_stats_t24 -- python3 tools/prove_many_timing.py 8 24 > $O/prove_stats_t24.log 2>&1)
cd $R && python3 tools/kconcurrency.py gpurun_out/final/prove_stats_t24 > $O/prove_concurrency_t24.txt
rm -f $O/prove_stats_t24/*/*kernel_trace.csv $O/msm_stats/*/*kernel_trace.csv $O/prove_stats/*/*kernel_trace.csv
echo all done >> $O/progress.log
