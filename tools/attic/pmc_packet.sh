#!/bin/bash
# Instruction-issue counters of the packet traversal kernel (last launch = cost-ordered).  usage: bash tools/pmc_packet.sh <tag>
set -u
TAG=${1:-p}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_packet_$TAG
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/trace_only.py --reps 4 --no-check > $OUT/g$i.log 2>&1
  grep "trace ms" $OUT/g$i.log | tail -1
done
cd $R
python3 - <<PY
import csv, glob, collections
last = {}
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_packet_kernel" in row["Kernel_Name"]:
            key = row["Counter_Name"]
            d = int(row["Dispatch_Id"])
            if key not in last or d >= last[key][0]:
                last[key] = (d, float(row["Counter_Value"]))
for k, v in sorted(last.items()):
    print(f"{k:28s} {v[1]:.6g}")
PY
