#!/usr/bin/env python3
"""Rebuild the same scene many times; every rebuild must give identical arrays and identical fast-mode hits
(race detector for the sort / refit protocols)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
with Context(0) as ctx:
    for name, tris, camz in (("rand4096", scenes.random_triangles(4096, seed=1), 300.0), ("rand70k", scenes.random_triangles(70000, seed=4), 300.0),
                             ("torus64k", scenes.tiled_torus(grid=2), 120.0), ("cfg2", scenes.tiled_torus(), 250.0)):
        d = RaytracingMeshDrawer(ctx, tris).awake()
        cam = scenes.camera(256, 192, (0.0, 0.0, camz))
        ref = None
        bad = {"keys": 0, "idx": 0, "bvh": 0, "fast": 0, "fast_vs_ref": 0}
        for r in range(reps):
            d.rebuild()
            c = d.container
            k = c.keys.get_data().copy(); i = c.triangle_index.get_data().copy(); b = c.bvh_data.get_data().copy()
            d.update(cam, mode=L.TRACE_FAST); f = d.hits()["t"].copy()
            d.update(cam, mode=L.TRACE_REFERENCE); rr = d.hits()["t"].copy()
            if ref is None:
                ref = (k, i, b, f)
            bad["keys"] += int((k != ref[0]).any()); bad["idx"] += int((i != ref[1]).any())
            n = len(tris)
            bad["bvh"] += int((b["min"][: n - 1] != ref[2]["min"][: n - 1]).any() or (b["max"][: n - 1] != ref[2]["max"][: n - 1]).any())
            bad["fast"] += int((f != ref[3]).any()); bad["fast_vs_ref"] += int((f != rr).any())
        print(name, len(tris), bad, flush=True)
        d.on_destroy()
