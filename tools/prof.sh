#!/bin/bash
# Runs on the GPU box: kernel-trace stats + the two PMC passes for bench.py (rocprofv3).
# usage: bash tools/prof.sh <tag>   (bench.py's own rocprofv3 child passes are switched off inside these runs: no nesting)
set -u
TAG=${1:-r1}
shift || true
EXTRA="$*"        # further bench.py arguments, e.g. --workload cfg4 --no-dynamic --no-sort-bench
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-counters --no-in-flight $EXTRA > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-counters --no-in-flight $EXTRA > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-counters --no-in-flight $EXTRA > $OUT/pmc_write.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
# keep the merge small: drop raw traces, keep stats + summary
find $OUT -name "*kernel_trace.csv" -size +8M -delete
