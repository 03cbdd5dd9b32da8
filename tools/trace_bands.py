"""Times the traversal per 8-row band and per 64-px column block to find where the time goes."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N  # noqa: E402
from unitysimpleraytracing_amd import layouts as L  # noqa: E402
from unitysimpleraytracing_amd import scenes  # noqa: E402
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer  # noqa: E402

W, H = 1920, 1080
tris = scenes.tiled_torus()
cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
ctx = Context(0)
d = RaytracingMeshDrawer(ctx, tris).awake()
hits = DataBuffer(ctx, W * H, L.HIT)
stats = DataBuffer(ctx, 1, L.TRACE_STATS)
s = d.container.scene()
e0, e1 = ctx.event(), ctx.event()


def t(rect, mode=L.TRACE_FAST, reps=3, st=False):
    best = 1e9
    for _ in range(reps):
        ctx.record(e0)
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), rect[0], rect[1], rect[2], rect[3], C.byref(s), mode,
                                                     hits.device, stats.device if st else None))
        ctx.record(e1)
        best = min(best, ctx.elapsed_ms(e0, e1))
    return best


print("full", t((0, 0, W, H)))
rows = [(y, t((0, y, W, min(y + 40, H)))) for y in range(0, H, 40)]
print("bands of 40 rows (ms):", " ".join(f"{y}:{ms:.3f}" for y, ms in rows))
worst = max(rows, key=lambda r: r[1])[0]
cols = [(x, t((x, worst, min(x + 64, W), min(worst + 40, H)))) for x in range(0, W, 64)]
print(f"band y={worst}, 64-px column blocks (ms):", " ".join(f"{x}:{ms:.3f}" for x, ms in cols))
wx = max(cols, key=lambda r: r[1])[0]
for yy in range(worst, min(worst + 40, H), 8):
    for xx in range(wx, min(wx + 64, W), 8):
        ms = t((xx, yy, xx + 8, yy + 8), st=True, reps=1)
        stv = stats.get_data()[0]
        print(f"tile ({xx},{yy}): {ms:.3f} ms pops={int(stv['pops'])} tri_tests={int(stv['tri_tests'])}")
ctx.close()
