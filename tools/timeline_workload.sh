#!/bin/bash
# one replayed iteration's kernel timeline of a bench workload; usage: timeline_workload.sh <workload> <batch> <tag>
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
d=gpurun_out/prof_$3
rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --workload $1 --no-cpu-baseline --no-secondary --no-extra --corr-n 0 --steps 20 --warmup 3 --batch $2 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
python3 tools/trace_timeline.py $(find $d -name '*kernel_trace.csv' | head -1) > gpurun_out/$3_timeline.txt
rm -rf $d
