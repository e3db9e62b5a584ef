# headline configuration under BPG_MSM_C (window bits); run on the GPU box from the repo root
set -e
O=gpurun_out/csweep
mkdir -p $O
for v in 14 15 16; do
  BPG_MSM_C=$v timeout -k 10 120 python3 bench.py --no-proofs --no-big-proof --no-batch --no-extras --logn ${LOGN:-20} > $O/c_$v.json 2> $O/c_$v.err
  python3 -c "
import json,sys
d=json.loads(open('$O/c_$v.json').read().strip().splitlines()[-1])
print('c', $v, 'value %.1f M/s' % (d['value']/1e6), 'ms/step %.3f' % d['ms_per_step'], 'acc avg %.1f alone %.1f' % (d['roofline']['kernel_avg_us'], d['roofline']['kernel_alone_us']), 'parity', d['cpu_baseline']['parity_with_gpu'])"
done
