# headline configuration under BPG_MSM_LS0 (entries per accumulate thread); run on the GPU box from the repo root
set -e
O=gpurun_out/ls0
mkdir -p $O
for v in ${LS0_LIST:-16 24 32 48 64}; do
  BPG_MSM_LS0=$v timeout -k 10 120 python3 bench.py --no-proofs --no-big-proof --no-batch --no-cpu-baseline --no-extras > $O/ls0_$v.json 2> $O/ls0_$v.err
  python3 -c "
import json,sys
d=json.loads(open('$O/ls0_$v.json').read().strip().splitlines()[-1])
print('ls0', $v, 'value %.1f M/s' % (d['value']/1e6), 'ms/step %.3f' % d['ms_per_step'], 'acc avg %.1f alone %.1f' % (d['roofline']['kernel_avg_us'], d['roofline']['kernel_alone_us']))"
done
