#!/usr/bin/env python3
"""cfg4 on one GPU, timing only: rebuild ms + per-kernel profile + the 1080p frame (tools/cfg4_single.py also checks it against the oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitysimpleraytracing_amd import layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer
tris = scenes.tiled_torus(nu=400, nv=160)
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, tris).awake()
    for _ in range(2):
        d.rebuild()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        d.rebuild()
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3 / 5
    ctx.profile_begin()
    for _ in range(3):
        d.rebuild()
    prof = {k: round(v[1] / 3, 4) for k, v in ctx.profile_end().items()}
    print("rebuild ms", round(ms, 3), "Mtri/s", round(len(tris) / ms / 1e3, 1), prof, flush=True)
    cam = scenes.camera(1920, 1080, (0.0, 0.0, 250.0))
    d.rebuild()
    for _ in range(3):
        d.update(cam, mode=L.TRACE_FAST)
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); d.update(cam, mode=L.TRACE_FAST); ctx.record(e1)
    print("trace ms", round(ctx.elapsed_ms(e0, e1), 3))
    d.on_destroy()
