#!/bin/bash
# phase stamps of the direct (skinny) kernel on the three W_1-streaming products of a 32-row critic step
# (stamps: start | first pass multiplied | all passes | stores issued | drained)
cd $GRAFT_REPO_ROOT
for c in "64 256 18161 0 0 -1 0" "32 18154 256 0 1 -1 0" "32 256 18154 0 0 -1 0"; do
  echo "=== $c"
  timeout -k 10 60 python3 tools/stamps.py $c 2>&1 | grep -v amdgpu.ids | head -24 || exit 1
done
