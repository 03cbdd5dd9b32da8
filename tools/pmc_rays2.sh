#!/bin/bash
# Issue + memory-path counters of the per-ray traversal kernel of the cfg5 bench (mean per launch = one bounce).  usage: bash tools/pmc_rays2.sh <tag>
# (every pass under its own `timeout`: an unknown counter name once made rocprofv3 sit on the box until gpurun's limit)
set -u
TAG=${1:-r}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_rays_$TAG
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TD_TD_BUSY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/dynamic_bench.py > $OUT/g$i.log 2>&1
  echo "group $i rc=$?: $(grep -o '"trace_rays[a-z_]*kernel": [0-9.]*' $OUT/g$i.log | tail -1)"
done
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_rays" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for k, v in sorted(acc.items()):
        line = f"{k:40s} launches={len(v)} mean per launch={sum(v)/len(v):.6g}"
        print(line); fh.write(line + "\n")
PY
rm -rf $OUT/g*/
