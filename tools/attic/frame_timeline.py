#!/usr/bin/env python3
"""Timeline of the LAST frame in a rocprofv3 kernel trace: start / end / duration / gap before, us from the frame's first kernel.

  rocprofv3 --kernel-trace -d gpurun_out/ft -o ft --output-format csv -- python3 tools/dynamic_bench.py --frames 3 --no-profile
  python3 tools/frame_timeline.py gpurun_out/ft [first-kernel-substring, default animate_morton]
"""
import csv, glob, sys
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "animate_morton"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if len(starts) < 2:
    sys.exit("need two frames in the trace")
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
end_prev = None
busy = 0
for s, e, name, q in rows[a:b]:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    gap = (s - end_prev) / 1e3 if end_prev is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  q{q}  {name}")
    end_prev = max(end_prev, e) if end_prev is not None else e
print(f"frame: {(rows[b][0] - t0) / 1e3:.1f} us from first kernel to the next frame's first kernel")
