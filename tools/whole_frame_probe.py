"""Static-camera whole frames (history-driven dispatch, no rebuild between them) of the cfg2 scene from both cameras, at 1080p
and 3840x2160, and of the cfg4 scene (16 M triangles) — the cases the whole-frame cooperative rule has to suit at once."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer

def frames(ctx, d, cam, w, h, reps=12):
    hits = DataBuffer(ctx, w * h, L.HIT)
    s = d.container.scene()
    c = N.Camera.from_dict(cam)
    e0, e1 = ctx.event(), ctx.event()
    best, tot = 1e9, 0.0
    for k in range(reps + 3):
        ctx.record(e0)
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(c), 0, 0, w, h, C.byref(s), L.TRACE_FAST, hits.device, None))
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)
        if k >= 3:
            best = min(best, ms); tot += ms
    hits.dispose()
    return tot / reps, best

with Context(0) as ctx:
    out = []
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
    for name, cam, w, h in (("cfg2 z=250 1080p", scenes.camera(1920, 1080, (0, 0, 250.0)), 1920, 1080),
                            ("cfg2 z=160 1080p", scenes.camera(1920, 1080, (0, 0, 160.0)), 1920, 1080),
                            ("cfg2 z=250 4K", scenes.camera(3840, 2160, (0, 0, 250.0)), 3840, 2160),
                            ("cfg2 z=160 4K", scenes.camera(3840, 2160, (0, 0, 160.0)), 3840, 2160)):
        m, b = frames(ctx, d, cam, w, h)
        out.append(f"{name}: {m * 1e3:.1f} us (best {b * 1e3:.1f})")
    d.on_destroy()
    if "--cfg4" in sys.argv:
        d = RaytracingMeshDrawer(ctx, scenes.tiled_torus(nu=400, nv=160)).awake()
        m, b = frames(ctx, d, scenes.camera(1920, 1080, (0, 0, 250.0)), 1920, 1080, reps=6)
        out.append(f"cfg4 z=250 1080p: {m * 1e3:.1f} us (best {b * 1e3:.1f})")
        d.on_destroy()
    print(" | ".join(out))
