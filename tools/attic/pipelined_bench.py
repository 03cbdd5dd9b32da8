#!/usr/bin/env python3
"""Two frames in flight on ONE GPU: two contexts (each with its own scene buffers, scratch and stream) take the steps in
turn, so step k+1's rebuild (latency / bandwidth bound) runs beside step k's trace (vector-issue bound).  Every step is
still a full rebuild followed by the trace of THAT rebuild's scene.  Prints ms per step for 1 and 2 contexts."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer

W, H = 1920, 1080
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tris = scenes.tiled_torus()
cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))


def run(n_ctx):
    ctxs = [Context(0) for _ in range(n_ctx)]
    drawers = [RaytracingMeshDrawer(c, tris).awake(fast=True) for c in ctxs]
    hits = [DataBuffer(c, W * H, L.HIT) for c in ctxs]

    def step(k):
        i = k % n_ctx
        drawers[i].rebuild(fast=True)
        s = drawers[i].container.scene()
        N.check(ctxs[i].handle, N.lib.lbvh_trace_primary(ctxs[i].handle, C.byref(cam), 0, 0, W, H, C.byref(s), L.TRACE_FAST, hits[i].device, None))
    for k in range(4 * n_ctx):
        step(k)
    for c in ctxs:
        c.sync()
    best = 1e9
    for rnd in range(3):
        t0 = time.perf_counter()
        for k in range(K):
            step(k)
        for c in ctxs:
            c.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / K)
    nh = [int((h.get_data()["t"] < 2.0e9).sum()) for h in hits]
    for d in drawers:
        d.on_destroy()
    for c in ctxs:
        c.close()
    return best, nh


for n in (1, 2, 3):
    ms, nh = run(n)
    print(f"{n} context(s): {ms:.4f} ms per step (wall clock, best of 3 x {K} steps), hits per frame {nh}")
