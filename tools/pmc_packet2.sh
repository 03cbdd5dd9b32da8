#!/bin/bash
# Issue + memory-path counters of the packet traversal kernel (last launch = cost-ordered).  usage: bash tools/pmc_packet2.sh <tag>
set -u
TAG=${1:-p}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_packet_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_FLAT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TD_TD_BUSY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/trace_only.py --reps 4 --no-check > $OUT/g$i.log 2>&1
  echo "group $i: $(grep 'trace ms' $OUT/g$i.log | tail -1) $(grep -i 'error\|invalid\|not found' $OUT/g$i.log | head -2)"
done
cd $R
python3 - <<PY
import csv, glob
last = {}
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_packet_kernel" in row["Kernel_Name"]:
            key = row["Counter_Name"]
            d = int(row["Dispatch_Id"])
            if key not in last or d > last[key][0]:
                last[key] = (d, float(row["Counter_Value"]))
            elif d == last[key][0]:
                last[key] = (d, last[key][1] + float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for k, v in sorted(last.items()):
        line = f"{k:36s} {v[1]:.6g}"
        print(line); fh.write(line + "\n")
PY
rm -rf $OUT/g*/
