#!/bin/bash
# The packet walk with its slab products on the matrix pipe against the vector-pipe form (LBVH_LIB=build_exp/liblbvh_valu.so,
# tools/build_variant.sh valu -DLBVH_LEAN_VALU_PRODUCTS): issue counters of the plain packet kernel on COLD frames of cfg2
# (history dropped before every frame: trace_packet_kernel, no cooperative tiles), mean per launch.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/r4f
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_]*MFMA[A-Z_]*" | sort -u > $OUT/mfma_counters.txt
for lib in matrix valu; do
  if [ $lib = valu ]; then export LBVH_LIB=$R/build_exp/liblbvh_valu.so; else unset LBVH_LIB; fi
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_CYCLES" \
             "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" ; do
    i=$((i+1))
    rm -rf /tmp/pm_$lib$i
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm_$lib$i -- python3 $R/tools/trace_only.py --reps 6 --no-check --cold > /tmp/pm_$lib$i.log 2>&1
  done
  python3 - $lib <<'PY'
import csv, glob, sys, collections
lib = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
dur = []
for f in glob.glob(f"/tmp/pm_{lib}*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "trace_packet_kernel" in r["Kernel_Name"]:
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, _), v in per.items():
        acc[name][0] += 1; acc[name][1] += v
for f in glob.glob(f"/tmp/pm_{lib}1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_packet_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"== {lib}: trace_packet_kernel, cold frames; kernel us (first counter pass): {['%.1f' % d for d in dur]}")
for k in sorted(acc):
    n, v = acc[k]
    print(f"   {k:34s} {v / n:14.6g}")
PY
done
