"""Round 5, VERDICT r4 item 2: cfg5 with the live rays SORTED between bounces (lbvh_debug_switch LBVH_DEBUG_RAY_SORT: 1 = direction
octant then origin Morton code, 2 = origin Morton code then octant) against the product's pixel-order list.  Per-kernel device
times of one frame's kernels (library profile; the experiment's form synchronises once per bounce to read the live count, so only
the kernels are compared, not the frame), and the image must be identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitysimpleraytracing_amd import _native as N  # noqa: E402
from unitysimpleraytracing_amd import scenes  # noqa: E402
from unitysimpleraytracing_amd.host import Context, DynamicPathTracer  # noqa: E402

ctx = Context(0)
tris, body, centres = scenes.tiled_torus(with_bodies=True)
cam = scenes.camera(1920, 1080, (0.0, 0.0, 250.0))
pt = DynamicPathTracer(ctx, tris, body, centres)
images = {}
for form in (0, 1, 2):
    ctx.debug_switch(N.DEBUG_SWITCH_RAY_SORT, form)
    for f in range(3):
        pt.animate(0.01 * f)
        pt.render(cam, 4)
    ctx.sync()
    ctx.profile_begin()
    for f in range(5):
        pt.animate(0.01 * f)
        pt.render(cam, 4)
    prof = ctx.profile_end()
    pt.animate(0.02)
    pt.render(cam, 4)
    images[form] = pt.image().copy()
    walk = sum(v[1] for k, v in prof.items() if "trace_rays_wide" in k) / 5.0
    sort = sum(v[1] for k, v in prof.items() if k.startswith("sort_") or "ray_sort_keys" in k) / 5.0
    build_sort = 0.0
    print(f"form {form}: walk kernels {walk:.4f} ms per frame, sort + key kernels {sort:.4f} ms (of which the rebuild's own sort ~0.058)")
    print("   " + ", ".join(f"{k.split('<')[0]} {v[1] / 5.0 * 1e3:.0f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]))
ctx.debug_switch(N.DEBUG_SWITCH_RAY_SORT, 0)
for form in (1, 2):
    same = (images[form].view(np.uint16) == images[0].view(np.uint16)).all()
    print(f"form {form}: image identical to the unsorted walk's: {bool(same)}")
ctx.close()
