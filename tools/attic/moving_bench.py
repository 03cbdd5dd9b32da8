#!/usr/bin/env python3
"""Traversal time of the cfg2 frame with a static camera, with the dispatch history dropped before every frame (cold)
and with the camera yawing / dollying between frames: what the previous frame's per-tile costs are worth as a
dispatch hint when the picture changes.  Prints mean ms per frame (HIP events around the trace)."""
import ctypes as C, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
from bench import yawed
W, H = 1920, 1080
SHARDS = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # trace shard 0 of SHARDS (one GPU's share of the frame)
EXCHANGE = len(sys.argv) > 2 and sys.argv[2] == "exchange"


def exchange_main():
    """One-GPU stand-in for SHARDS ranks that exchange their per-tile costs after every frame (lbvh_trace_costs_export /
    _import; on a real node: an all-reduce MAX of the 130 KB arrays while the next rebuild runs).  One context per rank,
    all on this GPU; every frame all shares are traced, only rank 0's is timed."""
    tris = scenes.tiled_torus()
    ctxs = [Context(0) for _ in range(SHARDS)]
    drawers = [RaytracingMeshDrawer(c, tris).awake() for c in ctxs]
    hits = [DataBuffer(c, W * H, L.HIT) for c in ctxs]
    scn = [d.container.scene() for d in drawers]
    tx, ty = (W + 7) // 8, (H + 7) // 8
    frame = DataBuffer(ctxs[0], tx * ty, np.uint32)
    base = scenes.camera(W, H, (0.0, 0.0, 250.0))

    def run(label, cams, exchange, frames=24):
        e0, e1 = ctxs[0].event(), ctxs[0].event()
        tot = 0.0
        for c in ctxs:
            c.trace_forget()
        for k in range(frames + 4):
            cam = N.Camera.from_dict(cams(k))
            for r, c in enumerate(ctxs):
                if r == 0:
                    c.record(e0)
                N.check(c.handle, N.lib.lbvh_trace_primary_shard(c.handle, C.byref(cam), r, SHARDS, C.byref(scn[r]), L.TRACE_FAST, hits[r].device, None))
                if r == 0:
                    c.record(e1)
                c.sync()
            if k >= 4:
                tot += ctxs[0].elapsed_ms(e0, e1)
            if exchange:
                frame.fill_u32(0, mirror=False)
                ctxs[0].sync()
                for c in ctxs:                       # disjoint tiles: every rank writes its own (= an all-reduce over ranks)
                    c.trace_costs_export(frame, tx, ty)
                    c.sync()
                for c in ctxs:
                    c.trace_costs_import(frame, tx, ty)
                    c.sync()
        print(f"{label:44s} {tot / frames:.4f} ms")
    tri = lambda k: abs((k % 16) - 8) - 4
    run("static", lambda k: base, False)
    for name, cams in (("yaw 0.25 deg/frame", lambda k: yawed(base, 0.25 * (k + 1))), ("yaw 1 deg/frame, +-4 deg", lambda k: yawed(base, 1.0 * tri(k))),
                       ("dolly 1 unit/frame", lambda k: scenes.camera(W, H, (0.0, 0.0, 250.0 - k))),
                       ("strafe 2 units/frame, +-8", lambda k: scenes.camera(W, H, (2.0 * tri(k), 0.0, 250.0)))):
        run(name + ", own costs only", cams, False)
        run(name + ", costs exchanged", cams, True)
    for d in drawers:
        d.on_destroy()
    for c in ctxs:
        c.close()


if EXCHANGE:
    print(f"share 1/{SHARDS} of the 1080p frame, rank 0 timed")
    exchange_main()
    sys.exit(0)

with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
    hits = DataBuffer(ctx, W * H, L.HIT)
    s = d.container.scene()
    base = scenes.camera(W, H, (0.0, 0.0, 250.0))

    def run(label, cams, forget=False, frames=24):
        e0, e1 = ctx.event(), ctx.event()
        tot = 0.0
        for k in range(frames + 4):
            if forget:
                ctx.trace_forget()
            cam = N.Camera.from_dict(cams(k))
            ctx.record(e0)
            N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(cam), 0, SHARDS, C.byref(s), L.TRACE_FAST, hits.device, None))
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1)
            if k >= 4:
                tot += ms
        print(f"{label:28s} {tot / frames:.4f} ms")
    run("static", lambda k: base)
    run("cold", lambda k: base, forget=True)
    for deg in (0.25, 1.0, 3.0):
        run(f"yaw {deg} deg/frame", lambda k: yawed(base, deg * (k + 1)))
    run("dolly 1 unit/frame", lambda k: scenes.camera(W, H, (0.0, 0.0, 250.0 - k)))
    # back and forth, so that the scene stays in view (the one-way turns above end up looking past it)
    tri = lambda k: abs((k % 16) - 8) - 4            # -4 .. 4 in steps of 1
    for deg in (1.0, 2.0):
        run(f"yaw {deg} deg/frame, +-{4 * deg:.0f} deg", lambda k: yawed(base, deg * tri(k)))
    run("strafe 2 units/frame, +-8", lambda k: scenes.camera(W, H, (2.0 * tri(k), 0.0, 250.0)))
    d.on_destroy()
