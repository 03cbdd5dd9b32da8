#!/bin/bash
# PMC passes over the traversal kernel only.  usage: bash tools/pmc_trace.sh <tag> [mode]
set -u
TAG=${1:-t}; MODE=${2:-fast}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
[ -f $R/gpurun_out/counters_list.txt ] || rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/trace_only.py --reps 2 --mode $MODE > $OUT/g$i.log 2>&1
  tail -2 $OUT/g$i.log
done
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:36s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
