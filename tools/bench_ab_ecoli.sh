#!/bin/bash
# bench-only A/B of library builds on the GTEx and E. coli batch-32 workloads; usage: bench_ab_ecoli.sh "<variants>" [repeats]
cd $GRAFT_REPO_ROOT
L=adversarial_gene_expression_amd/libagx.so
cp $L /tmp/libagx_tree.so
for rep in $(seq 1 ${2:-2}); do
for v in $1; do
  if [ $v = tree ]; then cp /tmp/libagx_tree.so $L; else cp tools/ab/libagx_$v.so $L; fi
  b32=$(timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  ec=$(timeout -k 10 120 python3 bench.py --workload ecoli --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "$v: gtex batch 32: $b32 | ecoli batch 32: $ec"
done
done
cp /tmp/libagx_tree.so $L
