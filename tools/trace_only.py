"""Builds cfg2 once and launches only the traversal kernel (for rocprofv3 counter passes)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N  # noqa: E402
from unitysimpleraytracing_amd import layouts as L  # noqa: E402
from unitysimpleraytracing_amd import scenes  # noqa: E402
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mode", default="fast")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--cam-z", type=float, default=250.0)
ap.add_argument("--cold", action="store_true", help="drop the dispatch history before every frame (the plain packet kernel, row-major)")
args = ap.parse_args()
W, H = 1920, 1080
tris = scenes.tiled_torus()
cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, args.cam_z)))
ctx = Context(0)
d = RaytracingMeshDrawer(ctx, tris).awake()
hits = DataBuffer(ctx, W * H, L.HIT)
s = d.container.scene()
mode = L.TRACE_FAST if args.mode == "fast" else L.TRACE_REFERENCE
import numpy as np
if args.no_check:
    e0, e1 = ctx.event(), ctx.event()
    for r in range(args.reps):
        if args.cold:
            ctx.trace_forget()
        ctx.record(e0)
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), mode, hits.device, None))
        ctx.record(e1)
        print("trace ms", ctx.elapsed_ms(e0, e1))
    ctx.close()
    sys.exit(0)
stats = DataBuffer(ctx, 1, L.TRACE_STATS)
hits.fill_u32(0x7FC00000, mirror=False)          # a tile left untraced must not pass on an earlier frame's values
N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), mode, hits.device, stats.device))
st = stats.get_data()[0]
print("stats", {k: int(st[k]) for k in st.dtype.names}, "visits/ray", float(st["pops"]) / (W * H))
fast = hits.get_data().copy()
N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), L.TRACE_REFERENCE, hits.device, None))
ref = hits.get_data().copy()
print("parity: t equal", bool((fast["t"] == ref["t"]).all()), "tri equal frac", float((fast["tri"] == ref["tri"]).mean()))
e0, e1 = ctx.event(), ctx.event()
for r in range(args.reps):
    ctx.record(e0)
    N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), mode, hits.device, None))
    ctx.record(e1)
    print("trace ms", ctx.elapsed_ms(e0, e1))
ctx.close()
