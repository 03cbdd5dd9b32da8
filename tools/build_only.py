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
