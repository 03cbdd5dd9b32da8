"""Static (history-driven) against cold (first) frames of the cfg2 scene from several camera distances, 1080p — a rule that makes
a frame WITH history slower than one without (as the cooperative set did on cfg4) shows up here."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer

W, H = 1920, 1080
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake(fast=True)
    s = d.container.scene()
    hits = DataBuffer(ctx, W * H, L.HIT)
    stats = DataBuffer(ctx, 1, L.TRACE_STATS)
    e0, e1 = ctx.event(), ctx.event()
    for z in (0.0, 30.0, 60.0, 110.0, 160.0, 250.0, 400.0, 800.0):
        cam = N.Camera.from_dict(scenes.camera(W, H, (3.0, 2.0, z)))
        def frame(st=None):
            ctx.record(e0)
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), L.TRACE_FAST, hits.device, st))
            ctx.record(e1)
            return ctx.elapsed_ms(e0, e1)
        ctx.trace_forget()
        cold = frame()
        static = min(frame() for _ in range(6))
        stats.fill_u32(0)
        frame(stats.device)
        st = stats.get_data()[0]
        print(f"camera z = {z:5.0f}: cold {cold * 1e3:7.1f} us, static {static * 1e3:7.1f} us, steps {int(st['pops']):8d}, hit fraction {int(st['hits']) / (W * H):.3f}", flush=True)
    d.on_destroy()
