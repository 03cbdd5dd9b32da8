"""Condenses rocprofv3 output (kernel trace + FETCH_SIZE / WRITE_SIZE PMC passes) into a per-(kernel,
grid) table and traffic.json: launches, average duration, HBM-side bytes per launch.  FETCH_SIZE is
doubled: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, section HBM); both counters
are in KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].strip()


def grid_of(row):
    for k in ("Grid_Size", "Grid_Size_X"):
        if k in row and row[k]:
            return int(float(row[k]))
    return 0


dur = defaultdict(lambda: [0, 0.0])
kt = find("stats", "*kernel_trace.csv")
if kt:
    for row in csv.DictReader(open(kt)):
        k = (short(row["Kernel_Name"]), grid_of(row))
        dur[k][0] += 1
        dur[k][1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3   # us


def pmc(sub, counter):
    out = defaultdict(lambda: [0, 0.0])
    f = find(sub, "*counter_collection.csv")
    if not f:
        return out
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != counter:
            continue
        k = (short(row["Kernel_Name"]), grid_of(row))
        out[k][0] += 1
        out[k][1] += float(row["Counter_Value"])
    return out


fetch = pmc("pmc_fetch", "FETCH_SIZE")
write = pmc("pmc_write", "WRITE_SIZE")
print(f"{'kernel':44s} {'grid':>10s} {'launches':>8s} {'avg_us':>10s} {'total_ms':>9s} {'fetch_MB(x2)':>13s} {'write_MB':>9s}")
traffic = {}
for k, (n, us) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    f = fetch.get(k)
    w = write.get(k)
    fb = 2 * f[1] / f[0] * 1024 if f and f[0] else None
    wb = w[1] / w[0] * 1024 if w and w[0] else None
    print(f"{k[0][:44]:44s} {k[1]:10d} {n:8d} {us / n:10.2f} {us / 1e3:9.3f} "
          f"{(f'{fb / 1e6:.2f}' if fb is not None else '-'):>13s} {(f'{wb / 1e6:.2f}' if wb is not None else '-'):>9s}")
    traffic[f"{k[0]}@{k[1]}"] = {"launches": n, "avg_us": round(us / n, 3), "fetch_bytes_x2": fb, "write_bytes": wb}
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
