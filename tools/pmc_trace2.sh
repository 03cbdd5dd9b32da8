#!/bin/bash
# second PMC set over the traversal kernel.  usage: bash tools/pmc_trace2.sh <tag>
set -u
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc2_$TAG
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_REQ SQC_TC_INST_REQ" \
           "SQ_LEVEL_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVES_LT_64 SQ_WAVES_EQ_64 SQ_INSTS_VSKIPPED SQ_ITEMS" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/trace_only.py --reps 2 --no-check > $OUT/g$i.log 2>&1
  grep "trace ms" $OUT/g$i.log | tail -1
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
