#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, KB, one pass each) of every kernel of the cfg2 rebuild, mean per launch.
# usage: bash tools/pmc_build_traffic.sh <tag>
set -u
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_build_traffic_$TAG
mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/tools/rebuild_only.py 8 > $OUT/$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for name in sorted(acc):
        line = name + "  " + "  ".join(f"{k} {sum(v)/len(v)/1024:.1f} MB (x{len(v)})" for k, v in sorted(acc[name].items()))
        print(line); fh.write(line + "\n")
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
