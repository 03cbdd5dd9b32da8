#!/usr/bin/env python3
"""Total node fetches (steps) of one cold whole cfg2 frame, from the exported per-tile costs: what sharing costs in lost pruning."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
W, H = 1920, 1080
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
    cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
    hits = DataBuffer(ctx, W * H, L.HIT)
    s = d.container.scene()
    tx, ty = (W + 7) // 8, (H + 7) // 8
    costs = DataBuffer(ctx, tx * ty, np.uint32)
    for label, forget in (("cold", True), ("warm", False), ("warm", False)):
        if forget:
            ctx.trace_forget()
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), L.TRACE_FAST, hits.device, None))
        costs.fill_u32(0, mirror=False)
        ctx.trace_costs_export(costs, tx, ty)
        c = costs.get_data()
        print(f"{label}: steps {int(c.sum())}  tiles>=64: {int((c >= 64).sum())}  >=128: {int((c >= 128).sum())}  >=256: {int((c >= 256).sum())}  max {int(c.max())}")
    d.on_destroy()
