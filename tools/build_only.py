#!/usr/bin/env python3
"""Per-kernel times of the rebuild only (cfg2 by default; argv[1] = nu, argv[2] = nv of the torus grid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import scenes
from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 80
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tris = scenes.tiled_torus(nu=nu, nv=nv)
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, tris).awake()
    for _ in range(3):
        d.rebuild()
    ctx.sync()
    for fast in (True, False):          # the rebuild as bench.py times it: back to back on the stream (graph replay)
        for _ in range(3):
            d.rebuild(fast=fast)
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(20):
            d.rebuild(fast=fast)
        ctx.record(e1)
        print(f"rebuild(fast={fast}): {ctx.elapsed_ms(e0, e1) / 20:.4f} ms")
    def per_kernel(label, fn):
        for _ in range(2):
            fn()
        ctx.sync()
        ctx.profile_begin()
        for _ in range(10):
            fn()
        prof = ctx.profile_end()
        print(f"-- {label} (one stream, kernels one after the other)")
        for k, v in prof.items():
            print(f"   {k:44s} {v[0] // 10:3d} x {v[1] / v[0] * 1e3:9.1f} us")
    per_kernel("reference arrays only: lbvh_build_scene without LBVH_BUILD_FAST_SCENE", lambda: d.rebuild(fast=False))
    per_kernel("derived scene only: lbvh_build_fast_scene", d.build_fast_scene)
    d.rebuild()
    print("-- both lanes (kernels of the two lanes overlap: durations under contention)")
    ctx.profile_begin()
    for _ in range(10):
        d.rebuild()
    prof = ctx.profile_end()
    tot = 0.0
    for k, v in prof.items():
        print(f"{k:44s} {v[0] // 10:3d} x {v[1] / v[0] * 1e3:9.1f} us")
        tot += v[1] / 10
    print(f"triangles {len(tris)}  build {tot:.4f} ms  {len(tris) / tot / 1e3:.1f} Mtri/s")
    d.on_destroy()
