#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes, as the guide prescribes) of tools/ubench/bucketcopy: do bucket-local passes stay on chip?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_bc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_bc_$c -- $R/tools/ubench/bucketcopy 26 > /dev/null 2>&1
  python3 - "$c" <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(f"/tmp/pmc_bc_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        a = rows.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in rows.items():
    mb = v / n * 1024 / 1e6 * (2 if c == "FETCH_SIZE" else 1)
    print(f"{c:11s} {k:40s} launches {n:3d}  mean per launch {mb:9.1f} MB{' (x2, gfx950)' if c == 'FETCH_SIZE' else ''}   algorithmic 536.9 MB read + 536.9 MB written")
PY
done
