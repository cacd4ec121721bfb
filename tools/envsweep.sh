#!/bin/bash
# effect of HIP runtime knobs on the captured iteration (batch 32)
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --corr-n 0 --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
}
run A=1 &&
run HIP_FORCE_DEV_KERNARG=1 &&
run HIP_FORCE_DEV_KERNARG=0 &&
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 &&
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 &&
run GPU_MAX_HW_QUEUES=1 &&
run HSA_ENABLE_INTERRUPT=0 &&
run DEBUG_HIP_GRAPH_BATCH_PACKETS=1
