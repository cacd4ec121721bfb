#!/bin/bash
# wide critic (20000 genes, 1024-wide): fused dW_1 + RMSprop on 128x128 tiles (AGX_OPT_BIG_TILES=1) against 64x64 (default)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 1 0; do
  if [ $v = 1 ]; then export AGX_OPT_BIG_TILES=1; else unset AGX_OPT_BIG_TILES; fi
  r=$(timeout -k 10 200 python3 -c "
import bench
r = bench.run_gpu('wide', 32, 50, 10, 1, 0, roofline=False)
print('%.1f samples/s %.4f ms' % (r['value'], r['ms_per_step']))" 2>/dev/null | tail -1) || exit 1
  echo "128x128 fused tiles=$v: $r"
done
done
unset AGX_OPT_BIG_TILES
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cgan.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
