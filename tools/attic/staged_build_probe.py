"""The rebuild as the reference's own call sequence issues it (MeshBufferContainer keys -> ComputeBufferSorter.Sort ->
DistributeKeys -> ConstructTree -> ConstructBVH -> derived scene: six C-ABI calls, what the re-hosted C# classes make) against
lbvh_build_scene (one call), per-kernel device times of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import scenes
from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer

with Context(0) as ctx:
    d = RaytracingMeshDrawer(ctx, scenes.tiled_torus()).awake(fast=True)
    for staged in (True, False):
        for _ in range(3):
            d.rebuild(fast=True, staged=staged)
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(10):
            d.rebuild(fast=True, staged=staged)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 10
        ctx.profile_begin()
        d.rebuild(fast=True, staged=staged)
        prof = ctx.profile_end()
        print(("staged calls" if staged else "lbvh_build_scene"), f"{ms:.4f} ms per rebuild;", ", ".join(f"{k.split('<')[0]} {v[1] * 1e3:.1f}" for k, v in prof.items()))
    d.on_destroy()
