#!/bin/bash
# Where the packet kernel's line fetches are served: L1 / L2 request and hit counters of the traversal kernels (cfg2 frame, static
# camera, history order), mean per launch.  usage: bash tools/pmc_trace_l2.sh <tag> [LBVH_LIB variant .so]
set -u
TAG=${1:-l2}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_l2_$TAG
mkdir -p $OUT
cd /tmp
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "FETCH_SIZE" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/trace_only.py --reps 4 --no-check > $OUT/g$i.log 2>&1
  echo "group $i rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_" in row["Kernel_Name"]:
            name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for name in sorted(acc):
        for k, v in sorted(acc[name].items()):
            line = f"{name:44s} {k:34s} launches={len(v):3d} mean={sum(v)/len(v):.5g}"
            print(line); fh.write(line + "\n")
PY
rm -rf $OUT/g*/
