import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, DynamicPathTracer
ctx = Context(0)
tris, body, centres = scenes.tiled_torus(with_bodies=True)
cam = scenes.camera(1920, 1080, (0.0, 0.0, 250.0))
pt = DynamicPathTracer(ctx, tris, body, centres)
for f in range(3):
    pt.animate(0.01 * f); pt.render(cam, 4)
h = ctx.handle; s = pt.drawer.container.scene(); ccam = N.Camera.from_dict(cam); count = 1920 * 1080
N.check(h, N.lib.lbvh_trace_primary(h, C.byref(ccam), 0, 0, 1920, 1080, C.byref(s), L.TRACE_FAST, pt.hits.device, None))
ctx.profile_begin(); N.check(h, N.lib.lbvh_path_first_bounce(h, C.byref(ccam), C.byref(s), pt.states.device, pt.hits.device, pt.seed, pt.albedo, pt.t_min)); print("first", {k: round(v[1] * 1e3, 1) for k, v in ctx.profile_end().items()})
for b in range(1, 4):
    ctx.profile_begin(); N.check(h, N.lib.lbvh_path_bounce(h, C.byref(s), pt.states.device, pt.hits.device, count, b, pt.seed, pt.albedo, pt.t_min)); print("bounce", b, {k: round(v[1] * 1e3, 1) for k, v in ctx.profile_end().items()})
ctx.profile_begin(); N.check(h, N.lib.lbvh_path_scatter(h, C.byref(s), pt.hits.device, count, 4, pt.seed, pt.albedo, pt.states.device)); print("last", {k: round(v[1] * 1e3, 1) for k, v in ctx.profile_end().items()})
ctx.close()
