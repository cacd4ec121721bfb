#!/bin/bash
# Evidence for profiles/: bench line, rocprofv3 kernel traces (batch 32 and 512), PMC traffic passes.
# usage (on the GPU box): bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -o pipefail
tag=$1
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
# 1. the contract line (default flags)
timeout -k 10 600 python3 bench.py > $O/${tag}_bench.json.log 2> $O/${tag}_bench.err || { tail -5 $O/${tag}_bench.err; exit 1; }
# 2. kernel traces of the same command (shortened), per batch
for b in 32 512; do
  d=$O/prof_${tag}_$b
  rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --no-cpu-baseline --no-secondary --no-extra --corr-n 0 --steps 40 --warmup 5 --batch $b > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  python3 tools/trace_timeline.py $(find $d -name '*kernel_trace.csv' | head -1) > $O/${tag}_timeline_b$b.txt
  cp $(find $d -name '*kernel_stats.csv' | head -1) $O/${tag}_b${b}_kernel_stats.csv
  rm -rf $d $d.log
done
# 3. HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate passes
for b in 32 512; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_${tag}_${c}_$b
    rm -rf $d
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 bench.py --no-secondary --no-extra --no-cpu-baseline --steps 2 --warmup 1 --batch $b --corr-n 0 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    cp $(find $d -name '*counter_collection.csv' | head -1) $O/${tag}_pmc_${c}_b$b.csv
    rm -rf $d $d.log
  done
done
# 4. MFMA pipe busy per kernel of the 512-row iteration (and the correlation): its own counter pass
d=$O/pmc_${tag}_mfma_512
rm -rf $d
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $d -o p -- python3 bench.py --no-secondary --no-extra --no-cpu-baseline --steps 3 --warmup 1 --batch 512 --corr-n 20000 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
python3 tools/pmc_summary.py mfma $(find $d -name '*counter_collection.csv' | head -1) > $O/${tag}_pmc_mfma_b512_corr.txt
rm -rf $d $d.log
for b in 32 512; do
  python3 tools/pmc_summary.py traffic $O/${tag}_pmc_FETCH_SIZE_b$b.csv $O/${tag}_pmc_WRITE_SIZE_b$b.csv > $O/${tag}_pmc_traffic_b$b.txt
done
echo collected
