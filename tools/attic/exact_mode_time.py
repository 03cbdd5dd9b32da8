import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
W, H = 1920, 1080
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
    cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
    hits = DataBuffer(ctx, W * H, L.HIT)
    s = d.container.scene()
    for mode, name in ((L.TRACE_FAST, "fast"), (getattr(L, "TRACE_FAST_EXACT", 1), "exact"), (L.TRACE_FAST, "fast"), (getattr(L, "TRACE_FAST_EXACT", 1), "exact")):
        e0, e1 = ctx.event(), ctx.event()
        for _ in range(4):
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), mode, hits.device, None))
        ctx.record(e0)
        for _ in range(20):
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), mode, hits.device, None))
        ctx.record(e1)
        print(name, round(ctx.elapsed_ms(e0, e1) / 20 * 1e3, 1), "us per frame")
    d.on_destroy()
