#!/bin/bash
# PMC passes over one GEMM shape (GPU box).  usage: pmc_case.sh <tag> M N K ta tb cfg sk
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=$O/pmc_${tag}_$i; rm -rf $d
  timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d $d -o p -- python3 tools/one_case.py "$@" 5 > $d.log 2>&1 || { tail -3 $d.log; continue; }
  python3 - "$(find $d -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:70]
    if 'gemm' not in k and 'splitk' not in k: continue
    per[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k, c in per.items():
    n = len(cnt[k])
    print(k, 'launches', n)
    for name, v in sorted(c.items()):
        print('   %-34s %16.0f per launch' % (name, v / n))
PY
  rm -rf $d $d.log
done
