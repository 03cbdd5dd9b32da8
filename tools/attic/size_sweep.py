"""Rebuild and static-camera frame times of the tiled-torus scene at several triangle counts (the forms the library switches between:
two-level sort below 2^21 pairs, merged launches up to 2 M triangles, replayed two-stream graph above) — looks for sizes that fall
between the tuned configurations."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer

W, H = 1920, 1080
with Context(0) as ctx:
    cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
    hits = DataBuffer(ctx, W * H, L.HIT)
    for nu, nv in ((20, 13), (40, 25), (57, 36), (80, 50), (102, 64), (113, 71), (116, 73), (160, 100), (226, 142)):
        tris = scenes.tiled_torus(nu=nu, nv=nv)
        d = RaytracingMeshDrawer(ctx, tris).awake(fast=True)
        for _ in range(3):
            d.rebuild(fast=True)
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(10):
            d.rebuild(fast=True)
        ctx.record(e1)
        build_ms = ctx.elapsed_ms(e0, e1) / 10
        s = d.container.scene()
        best = 1e9
        for k in range(8):
            ctx.record(e0)
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), L.TRACE_FAST, hits.device, None))
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1)
            if k >= 3:
                best = min(best, ms)
        ctx.trace_forget()
        ctx.record(e0)
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), L.TRACE_FAST, hits.device, None))
        ctx.record(e1)
        cold = ctx.elapsed_ms(e0, e1)
        n = len(tris)
        print(f"{n:9d} triangles: rebuild {build_ms:.4f} ms = {n / build_ms / 1e3:7.1f} Mtri/s | frame static {best:.4f} ms, cold {cold:.4f} ms", flush=True)
        d.on_destroy()
