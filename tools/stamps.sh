#!/bin/bash
# per-workgroup phase timelines (diagnostic build tools/ab/libagx_stamps.so): three GEMM shapes of the 512-row step,
# the main-loop ablations of the 128x128 TN kernel, and the fused dW_1 + RMSprop launch inside the 32-row iteration
cd $GRAFT_REPO_ROOT
for c in "18161 256 96 1 0 -1 0" "512 18154 256 0 1 -1 0" "3072 18154 256 0 0 -1 0"; do
  echo "=== $c"
  timeout -k 10 60 python3 tools/stamps.py $c 2>&1 | grep -v amdgpu.ids | head -30 || exit 1
done
for cfg in 0 21 22 23; do
  echo "=== TN 3072x18154x256 cfg $cfg  (21: no global loads in the loop, 22: no LDS refill / barrier, 23: neither)"
  timeout -k 10 60 python3 tools/stamps.py 3072 18154 256 1 0 $cfg 1 2>&1 | grep -v amdgpu.ids | head -22 || exit 1
done
echo "=== fused dW_1 + RMSprop inside the iteration, 32 rows (M 18161, K 96)"
timeout -k 10 100 python3 tools/stamps_iter.py 32 18161 96 2>&1 | grep -v amdgpu.ids | head -26
