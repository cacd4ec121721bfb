#!/bin/bash
# A/B of an on/off environment flag of libagx on one box; usage: env_flag_ab.sh VAR [repeats]   (off = VAR unset, on = VAR=1)
cd $GRAFT_REPO_ROOT
run() {
  b32=$(timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  ec=$(timeout -k 10 120 python3 bench.py --workload ecoli --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "$1: gtex batch 32: $b32 | ecoli batch 32: $ec"
}
for rep in $(seq 1 ${2:-2}); do
  unset $1; run "$1 unset"
  export $1=1; run "$1=1"
done
