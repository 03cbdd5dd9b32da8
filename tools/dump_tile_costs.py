#!/usr/bin/env python3
"""Per-tile step counts of the cfg2 frame (plain one-wave walk) for a few cameras -> gpurun_out/r4b/tile_costs.npz (analysis of cost predictors on the CPU)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
from bench import yawed
W, H = 1920, 1080
out = {}
with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
    hits = DataBuffer(ctx, W * H, L.HIT)
    s = d.container.scene()
    tx, ty = (W + 7) // 8, (H + 7) // 8
    costs = DataBuffer(ctx, tx * ty, np.uint32)
    cams = {"z250": scenes.camera(W, H, (0.0, 0.0, 250.0)), "z250_yaw10": yawed(scenes.camera(W, H, (0.0, 0.0, 250.0)), 10.0),
            "z160": scenes.camera(W, H, (0.0, 0.0, 160.0)), "inside": scenes.camera(W, H, (3.0, 2.0, 1.0))}
    for name, cam in cams.items():
        ccam = N.Camera.from_dict(cam)
        ctx.trace_forget()
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s), L.TRACE_FAST, hits.device, None))
        costs.fill_u32(0, mirror=False)
        ctx.trace_costs_export(costs, tx, ty)
        out[name] = costs.get_data().reshape(ty, tx).copy()
        out[name + "_cam"] = np.asarray(cam["camera_to_world"], dtype=np.float32)
        print(name, int(out[name].sum()), int(out[name].max()))
    d.on_destroy()
os.makedirs("gpurun_out/r4b", exist_ok=True)
np.savez_compressed("gpurun_out/r4b/tile_costs.npz", **out)
