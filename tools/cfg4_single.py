#!/usr/bin/env python3
"""cfg4 on ONE GPU: 16M-triangle tiled torus (400x160 quads x 125 tiles), full build checked against the oracle,
1080p frame traced in both modes.  Prints timings."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
from unitysimpleraytracing_amd import layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer

t0 = time.perf_counter()
tris = scenes.tiled_torus(nu=400, nv=160)
print("gen", len(tris), round(time.perf_counter() - t0, 1), "s", flush=True)
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, tris).awake()
    ctx.sync()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(3):
        d.rebuild()
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    prof = {k: round(v[1] / 3, 4) for k, v in ctx.profile_end().items()}
    print("rebuild ms", round(ms, 3), "Mtri/s", round(len(tris) / ms / 1e3, 1), prof, flush=True)
    c = d.container
    t0 = time.perf_counter()
    b = O.Built(tris, capacity=c.capacity, threads=O.num_threads())
    print("oracle build", round(time.perf_counter() - t0, 1), "s", flush=True)
    bad_leaf, bad_inner = c.get_all_gpu_data()
    n = b.n
    ok = {"keys": bool((c.keys.local == b.keys).all()), "idx": bool((c.triangle_index.local == b.indices).all()),
          "internal": bool((c.bvh_internal_node.local.view(np.uint32) == b.internal.view(np.uint32)).all()),
          "leaf": bool((c.bvh_leaf_node.local.view(np.uint32) == b.leaf.view(np.uint32)).all()),
          "bvh_min": bool((c.bvh_data.local["min"][: n - 1] == b.bvh["min"][: n - 1]).all()),
          "bvh_max": bool((c.bvh_data.local["max"][: n - 1] == b.bvh["max"][: n - 1]).all()),
          "bad": [len(bad_leaf), len(bad_inner)]}
    print(ok, flush=True)
    cam = scenes.camera(1920, 1080, (0.0, 0.0, 250.0))
    for mode in (L.TRACE_FAST, L.TRACE_REFERENCE):
        d.update(cam, mode=mode); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); d.update(cam, mode=mode); ctx.record(e1)
        print("trace mode", mode, round(ctx.elapsed_ms(e0, e1), 3), "ms", flush=True)
        h = d.hits()
        if mode == L.TRACE_FAST:
            hf = h.copy()
    print("t equal", bool((hf["t"] == h["t"]).all()), "hit fraction", float((h["t"] < L.MAX_FLOAT).mean()))
    oh, _ = O.trace_primary(b, cam, step=(8, 8), threads=O.num_threads())
    print("oracle sample t equal", bool((oh["t"] == h["t"][::8, ::8][: oh.shape[0], : oh.shape[1]]).all()))
    d.on_destroy()
