#!/bin/bash
# direct-kernel variants (tools/ab/libagx_<v>.so) on the W_1-streaming shapes of a 32-row step, then the bench
cd $GRAFT_REPO_ROOT
L=adversarial_gene_expression_amd/libagx.so
cp $L /tmp/libagx_tree.so
for v in $1; do
  if [ $v = tree ]; then cp /tmp/libagx_tree.so $L; else cp tools/ab/libagx_$v.so $L; fi
  echo "== variant $v"
  for c in "64 256 18161 0 0 -1 0" "32 18154 256 0 1 -1 0" "32 256 18154 0 0 -1 0" "32 256 18154 0 1 -1 0" "64 1024 20007 0 0 -1 0" "32 20000 1024 0 1 -1 0" "1024 256 18161 0 0 3 0"; do
    timeout -k 10 60 python3 tools/one_case.py $c 50 2>/dev/null | tail -1 || exit 1
  done
done
cp /tmp/libagx_tree.so $L
