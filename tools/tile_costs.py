"""Per-tile packet cost distribution of the cfg2 frame + offline schedule simulation (in-order vs
longest-first over W concurrent waves, cost = node fetches)."""
import ctypes as C
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N  # noqa: E402
from unitysimpleraytracing_amd import layouts as L  # noqa: E402
from unitysimpleraytracing_amd import scenes  # noqa: E402
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer  # noqa: E402

W, H = 1920, 1080
tris = scenes.tiled_torus()
cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
ctx = Context(0)
d = RaytracingMeshDrawer(ctx, tris).awake()
hits = DataBuffer(ctx, W * H, L.HIT)
stats = DataBuffer(ctx, 1, L.TRACE_STATS)
tx, ty = (W + 7) // 8, (H + 7) // 8
costs = DataBuffer(ctx, tx * ty, np.uint32)
s = d.container.scene()
N.check(ctx.handle, N.lib.lbvh_trace_tile_costs(ctx.handle, C.byref(cam), C.byref(s), hits.device, stats.device, costs.device))
c = costs.get_data().astype(np.int64)
print("tiles", c.size, "sum", int(c.sum()), "mean", c.mean(), "median", np.median(c), "p90", np.percentile(c, 90),
      "p99", np.percentile(c, 99), "max", c.max())
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "tile_costs.npy"), c.reshape(ty, tx))


def simulate(order, waves):
    heap = [0] * waves
    heapq.heapify(heap)
    end = 0
    for k in order:
        t = heapq.heappop(heap) + c[k] + 8          # + ~8 steps of hand-over cost
        end = max(end, t)
        heapq.heappush(heap, t)
    return end


for waves in (2048, 7168):
    ideal = c.sum() / waves
    print(f"waves={waves}: ideal {ideal:.0f} steps; in-order {simulate(range(c.size), waves)}; "
          f"longest-first {simulate(np.argsort(-c), waves)}; max tile {c.max()}")
ctx.close()
