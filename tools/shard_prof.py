#!/usr/bin/env python3
"""Per-kernel times of one rank's share (shard 0 of N) of the cfg2 frame."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
W, H = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
    cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
    hits = DataBuffer(ctx, W * H, L.HIT)
    s = d.container.scene()
    for _ in range(4):
        N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(cam), 0, n, C.byref(s), L.TRACE_FAST, hits.device, None))
    ctx.sync()
    ctx.profile_begin()
    for _ in range(5):
        N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(cam), 0, n, C.byref(s), L.TRACE_FAST, hits.device, None))
    for k, v in ctx.profile_end().items():
        print(f"{k:44s} {v[0] // 5:3d} x {v[1] / v[0] * 1e3:9.1f} us")
    d.on_destroy()
