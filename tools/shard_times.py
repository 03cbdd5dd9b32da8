#!/usr/bin/env python3
"""Time of ONE rank's share of the cfg2 frame for N = 1, 2, 4, 8 ranks, on one GPU: what each of N GPUs would spend
in lbvh_trace_primary_shard (the BVH is replicated, every rank has its own GPU).  Without arguments: 1920x1080 (the metric's
frame) and 3840x2160 (a 1/8 share = a whole 1080p frame's worth of rays: where the latency floor of a share stops mattering)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
SIZES = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(1920, 1080), (3840, 2160)]
with Context(0) as ctx:
  d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
  for W, H in SIZES:
    print(f"-- {W}x{H}")
    cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
    hits = DataBuffer(ctx, W * H, L.HIT)
    s = d.container.scene()
    for n in (1, 2, 4, 8):
        worst = 0.0
        for r in range(n):
            e0, e1 = ctx.event(), ctx.event()
            best = 1e9
            for k in range(6):
                ctx.record(e0)
                N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(cam), r, n, C.byref(s), L.TRACE_FAST, hits.device, None))
                ctx.record(e1)
                ms = ctx.elapsed_ms(e0, e1)
                if k >= 2:
                    best = min(best, ms)
            worst = max(worst, best)
        print(f"N={n}: slowest rank {worst * 1e3:7.1f} us -> {W * H / worst / 1e3:9.1f} Mrays/s, efficiency vs N=1 x N: ", end="")
        if n == 1:
            base = W * H / worst
        print(f"{W * H / worst / (base * n):.2f}")
    hits.dispose()
  d.on_destroy()
