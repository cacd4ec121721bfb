#!/bin/bash
# column-split critic tails (AGX_TAIL_NSPLIT=1 = one workgroup per row group as before; unset = automatic)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 1 auto 2; do
  if [ $v = auto ]; then unset AGX_TAIL_NSPLIT; else export AGX_TAIL_NSPLIT=$v; fi
  b32=$(timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms  tails %.1f + %.1f us/iter' % (d['value'], d['ms_per_step'], d['kernel_us_per_iter'].get('disc_fwd_hidden',0), d['kernel_us_per_iter'].get('gp2_hidden',0)))") || exit 1
  b512=$(timeout -k 10 120 python3 bench.py --batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "AGX_TAIL_NSPLIT=$v: batch 32: $b32 | batch 512: $b512"
done
done
unset AGX_TAIL_NSPLIT
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_cgan.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_kernels.py tests/test_gpu_dp2.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4
