"""How long does ONE heavy 8x8 tile take alone on the chip, cold and warm?  (the frame's critical path)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
W, H = 1920, 1080
ctx = Context(0)
d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake()
cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
hits = DataBuffer(ctx, W * H, L.HIT)
stats = DataBuffer(ctx, 1, L.TRACE_STATS)
tx, ty = W // 8, H // 8
costs = DataBuffer(ctx, tx * ty, np.uint32)
s = d.container.scene()
N.check(ctx.handle, N.lib.lbvh_trace_tile_costs(ctx.handle, C.byref(cam), C.byref(s), hits.device, stats.device, costs.device))
c = costs.get_data().astype(np.int64).reshape(ty, tx)
order = np.argsort(-c.ravel())
big = DataBuffer(ctx, 64 << 20, np.uint32)          # 256 MB: evicts L2 and most of the Infinity Cache
e0, e1 = ctx.event(), ctx.event()
for rank in (0, 1, 5, 50, 500, 5000):
    t = order[rank]; y, x = divmod(int(t), tx)
    res = []
    for mode in ("cold", "warm", "warm", "warm"):
        if mode == "cold":
            big.fill_u32(rank, mirror=False)
        ctx.trace_forget()
        if len(sys.argv) > 1:
            ctx.clock_probe()            # a full-chip vector-ALU kernel right before: the clock is up when the lone wave starts
        ctx.record(e0)
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(cam), x * 8, y * 8, x * 8 + 8, y * 8 + 8, C.byref(s), L.TRACE_FAST, hits.device, None))
        ctx.record(e1)
        res.append(ctx.elapsed_ms(e0, e1) * 1e3)
    print(f"tile rank {rank}: {c[y, x]} steps; one wave alone: cold {res[0]:.1f} us, warm {res[1]:.1f} {res[2]:.1f} {res[3]:.1f} us "
          f"-> {res[0] / c[y, x] * 1e3:.0f} / {min(res[1:]) / c[y, x] * 1e3:.0f} ns per step")
ctx.close()
