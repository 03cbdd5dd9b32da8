#!/usr/bin/env python3
"""The cfg2 frame in one table: whole frame static / cold (dispatch history dropped before every frame) / camera yawing 1 degree
per frame, and one rank's share of the frame for N = 2, 4, 8 (static and cold) — mean (and worst) us of HIP events around the
trace alone.  `--check`: every frame also compared with the reference mode's t.  A library variant is measured by
LBVH_LIB=build_exp/liblbvh_<name>.so (tools/build_variant.sh), alternating with the product in one gpurun call."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
from bench import yawed
W, H = 1920, 1080
check = "--check" in sys.argv
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
    base = scenes.camera(W, H, (0.0, 0.0, 250.0))
    hits = DataBuffer(ctx, W * H, L.HIT)
    s = d.container.scene()
    ref = None
    if check:
        cam = N.Camera.from_dict(base)
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), 0, 0, W, H, C.byref(s), L.TRACE_REFERENCE, hits.device, None))
        ref = hits.get_data().copy()

    def run(label, shards, cams, forget, frames=16):
        e0, e1 = ctx.event(), ctx.event()
        tot, worst = 0.0, 0.0
        ctx.trace_forget()
        for k in range(frames + 3):
            cam = N.Camera.from_dict(cams(k))
            if forget:
                ctx.trace_forget()
            if check:
                hits.fill_u32(0x7FC00000, mirror=False)
            ctx.record(e0)
            N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(cam), 0, shards, C.byref(s), L.TRACE_FAST, hits.device, None))
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1)
            if k >= 3:
                tot += ms
                worst = max(worst, ms)
            if check and shards == 1 and cams(k) is base:
                got = hits.get_data()
                assert (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all(), label
                assert np.count_nonzero(got["tri"] != ref["tri"]) < 64, label
        print(f"{label:34s} {tot / frames * 1e3:8.1f} us   (worst {worst * 1e3:7.1f})", flush=True)

    run("whole static", 1, lambda k: base, False)
    run("whole cold", 1, lambda k: base, True)
    run("whole yaw 1 deg/frame", 1, lambda k: yawed(base, 1.0 * (k + 1)), False)
    for n in (2, 4, 8):
        run(f"1/{n} share static", n, lambda k: base, False)
        run(f"1/{n} share cold", n, lambda k: base, True)
    d.on_destroy()
