#!/bin/bash
# rocprofv3 kernel trace of the batch-32 bench + timeline of one replayed iteration
# usage: bash tools/prof_timeline.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --no-cpu-baseline --no-secondary --corr-n 0 --steps 40 --warmup 5 "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
f=$(find $out -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py $f > gpurun_out/timeline_$tag.txt
cp $(find $out -name '*kernel_stats.csv' | head -1) gpurun_out/kernel_stats_$tag.csv
rm -rf $out
tail -22 gpurun_out/timeline_$tag.txt
