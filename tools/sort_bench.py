"""Sort-only benchmark: per-kernel device ms and G keys/s at several sizes (random 32-bit keys)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N  # noqa: E402
from unitysimpleraytracing_amd.host import Context, DataBuffer  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [20, 24, 26]
ctx = Context(0)
for lg in sizes:
    n = 1 << lg
    rng = np.random.default_rng(3)
    keys = DataBuffer(ctx, n, np.uint32)
    vals = DataBuffer(ctx, n, np.uint32)
    keys.local[:] = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    vals.local[:] = np.arange(n, dtype=np.uint32)
    best = 1e9
    prof = {}
    for r in range(4):
        keys.sync(); vals.sync()
        e0, e1 = ctx.event(), ctx.event()
        if r == 3:
            ctx.profile_begin()
        ctx.record(e0)
        N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, keys.device, vals.device, n))
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)
        if r == 3:
            prof = ctx.profile_end()
        elif r > 0:
            best = min(best, ms)
    k = keys.get_data()
    assert (k[1:] >= k[:-1]).all()
    per = {name: round(v[1] / v[0], 4) for name, v in prof.items()}
    print(f"2^{lg}: {best:.3f} ms  {n / best / 1e6:.2f} Gkeys/s  per-launch ms {per}")
    keys.dispose(); vals.dispose()
ctx.close()
