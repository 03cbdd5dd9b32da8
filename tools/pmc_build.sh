#!/bin/bash
# Issue counters of the rebuild's kernels (tools/build_only.py under rocprofv3 --pmc).  usage: bash tools/pmc_build.sh <tag>
set -u
TAG=${1:-b}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_build_$TAG
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/build_only.py > $OUT/g$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    if "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c: continue
    valu_frac = c["SQ_INSTS_VALU"] * 4 / 1024 / max(c["GRBM_GUI_ACTIVE"], 1)
    util = c.get("SQ_THREAD_CYCLES_VALU", 0) / max(c.get("SQ_ACTIVE_INST_VALU", 1), 1) / 64
    wait = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k[:40]:40s} waves {c['SQ_WAVES']:8.0f} valu {c['SQ_INSTS_VALU']:.3g} salu {c['SQ_INSTS_SALU']:.3g} vmem {c['SQ_INSTS_VMEM_RD']:.3g} lds {c['SQ_INSTS_LDS']:.3g} cycles {c['GRBM_GUI_ACTIVE']:.3g} valu_issue {valu_frac:.2f} lane_util {util:.2f} waiting {wait:.2f}")
PY
