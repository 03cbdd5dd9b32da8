#!/bin/bash
# Issue / LDS / memory counters of every kernel of the cfg2 rebuild, mean per launch.  usage: bash tools/pmc_build.sh <tag>
# (every pass under its own `timeout`)
set -u
TAG=${1:-b}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_build_$TAG
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/rebuild_only.py 8 > $OUT/g$i.log 2>&1
  echo "group $i rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for name in sorted(acc):
        fh.write(name + "\n"); print(name)
        for k, v in sorted(acc[name].items()):
            line = f"    {k:36s} launches={len(v):4d} mean={sum(v)/len(v):.6g}"
            print(line); fh.write(line + "\n")
PY
rm -rf $OUT/g*/
