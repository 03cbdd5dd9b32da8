#!/usr/bin/env python3
"""cfg5 (BASELINE configs[4]): 1M-triangle dynamic scene — per frame: rigid rotation of the 125 bodies, Morton regen,
full LBVH rebuild, primary rays + 4 diffuse bounces at 1 spp, 1920x1080.  Prints one JSON line with the per-kernel
breakdown (library's own event-bracketed profile)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ctx, frames=20, warmup=3, bounces=4, w=1920, h=1080, profile=True):
    from unitysimpleraytracing_amd import scenes
    from unitysimpleraytracing_amd.host import DynamicPathTracer
    tris, body, centres = scenes.tiled_torus(with_bodies=True)
    cam = scenes.camera(w, h, (0.0, 0.0, 250.0))
    pt = DynamicPathTracer(ctx, tris, body, centres)
    for f in range(warmup):
        pt.animate(0.01 * f)
        pt.render(cam, bounces)
    ctx.sync()
    t0 = time.perf_counter()
    for f in range(frames):
        pt.animate(0.01 * (warmup + f))
        pt.render(cam, bounces)
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    out = {"workload": "cfg5: 1M-triangle dynamic scene, animate + full rebuild + primary + %d bounces, %dx%d, 1 spp" % (bounces, w, h),
           "ms_per_frame": round(ms, 3), "frames_per_s": round(1e3 / ms, 1),
           "Mrays_s_all_segments": round(w * h * (bounces + 1) / (ms * 1e-3) / 1e6, 1)}
    if profile:
        ctx.profile_begin()
        for f in range(5):
            pt.animate(0.01 * f)
            pt.render(cam, bounces)
        out["kernels_ms_per_frame"] = {k: round(v[1] / 5.0, 4) for k, v in ctx.profile_end().items()}
    img = pt.image()
    out["alpha_fraction"] = round(float((img[..., 3] > 0).mean()), 4)
    out["mean_rgb"] = [round(float(x), 4) for x in img[..., :3].astype(np.float32).mean(axis=(0, 1))]
    pt.drawer.on_destroy()
    return out


if __name__ == "__main__":
    from unitysimpleraytracing_amd.host import Context
    with Context(0) as ctx:
        print(json.dumps(run(ctx)))
