#!/bin/bash
# A/B of library builds on the GPU box: tools/ab/libagx_<variant>.so (built here with different -D options of
# csrc/gemm.hip, see the Makefile line in DESIGN.md section 6) are copied over the package's libagx.so in turn;
# single GEMM shapes of the 512- and 32-row steps, then the bench at both batch sizes.
#   usage: lib_ab.sh "<variant> <variant> ..."      (run through gpurun; the tree on the box is a scratch copy)
cd $GRAFT_REPO_ROOT
L=adversarial_gene_expression_amd/libagx.so
cp $L /tmp/libagx_tree.so
for v in $1; do
  if [ $v = tree ]; then cp /tmp/libagx_tree.so $L; else cp tools/ab/libagx_$v.so $L; fi
  echo "== variant $v"
  for c in "3072 18154 256 0 0 -1 0" "512 18154 256 0 1 -1 0" "1536 256 18161 0 0 -1 0" "512 256 18154 0 0 -1 0" "18161 256 1536 1 0 -1 0" "8192 8192 256 0 1 0 1" "4096 4096 4096 0 1 0 1" "18161 256 96 1 0 -1 0" "32 18154 256 0 1 -1 0"; do
    timeout -k 10 60 python3 tools/one_case.py $c 30 2>/dev/null | tail -1 || exit 1
  done
  timeout -k 10 120 python3 bench.py --batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench512', d['value'], d['ms_per_step'])" || exit 1
  timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench32', d['value'], d['ms_per_step'])" || exit 1
done
cp /tmp/libagx_tree.so $L
