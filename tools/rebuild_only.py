#!/usr/bin/env python3
"""The cfg2 rebuild alone (1 M-triangle tiled torus, reference + derived arrays), N times: the workload of tools/pmc_build.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitysimpleraytracing_amd import scenes
from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tris = scenes.tiled_torus()
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, tris).awake()
    for _ in range(reps):
        d.rebuild()
    ctx.sync()
    d.on_destroy()
