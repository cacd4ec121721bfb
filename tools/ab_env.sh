#!/bin/bash
# A/B of one environment knob on the batch-32 bench line (GPU box).  usage: ab_env.sh VAR [reps]
var=$1; reps=${2:-2}
cd $GRAFT_REPO_ROOT
for r in $(seq $reps); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $var=1; else unset $var; fi
    timeout -k 10 120 python3 -u bench.py --no-cpu-baseline --no-secondary --no-extra --corr-n 0 --steps 400 2>gpurun_out/ab.err > gpurun_out/ab.json || { tail -3 gpurun_out/ab.err; exit 1; }
    python3 -c "
import json
d = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
k = d['kernel_us_per_iter']
print('$var=$v', round(d['value']), 'samples/s', round(d['ms_per_step'] * 1e3, 1), 'us/iter', k)
"
  done
done
