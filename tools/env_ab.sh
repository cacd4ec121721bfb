#!/bin/bash
# A/B of an environment knob of libagx on one box; usage: env_ab.sh VAR "<value> <value> ..." [repeats]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${3:-2}); do
for v in $2; do
  export $1=$v
  b32=$(timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  b512=$(timeout -k 10 120 python3 bench.py --batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "$1=$v: batch 32: $b32 | batch 512: $b512"
done
done
