#!/bin/bash
# bench line under several environment settings (GPU box).  usage: ab_env2.sh "A=1 B=2" "C=1" ...   ("-" = no setting)
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  ( if [ "$setting" != "-" ]; then export $setting; fi
    timeout -k 10 120 python3 -u bench.py --no-cpu-baseline --no-secondary --no-extra --corr-n 0 --steps 400 2>gpurun_out/ab.err > gpurun_out/ab.json || { tail -3 gpurun_out/ab.err; exit 1; }
    python3 -c "
import json
d = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
k = d['kernel_us_per_iter']
print('[$setting]', round(d['value']), 'samples/s', round(d['ms_per_step'] * 1e3, 1), 'us/iter', 'dw_in', k.get('disc_dw_in'), 'gen_dw_out', k.get('gen_dw_out'), 'opt', k.get('opt_rmsprop'))
" )
done
