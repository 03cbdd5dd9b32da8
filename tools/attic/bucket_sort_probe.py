"""The two-level sort's bucket kernel against bucket size: random keys whose top byte takes `buckets` values, so that every
used bucket holds n / buckets pairs (+- a few %).  Prints the per-kernel device times of the forced two-level form and of the
four passes.  (round 5)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitysimpleraytracing_amd import _native as N  # noqa: E402
from unitysimpleraytracing_amd.host import Context, DataBuffer  # noqa: E402

ctx = Context(0)
cases = [(1 << 18, 256), (1 << 19, 256), (1 << 20, 256), (1 << 20, 128), (1 << 20, 85), (1 << 20, 64), (1 << 19, 64), (1 << 17, 256)]
for n, buckets in cases:
    rng = np.random.default_rng(3)
    k = rng.integers(0, 1 << 24, size=n, dtype=np.uint64) | (rng.integers(0, buckets, size=n, dtype=np.uint64) << 24)
    keys = DataBuffer(ctx, n, np.uint32)
    vals = DataBuffer(ctx, n, np.uint32)
    keys.local[:] = k.astype(np.uint32)
    vals.local[:] = np.arange(n, dtype=np.uint32)
    line = f"n = 2^{int(np.log2(n))}, {buckets} buckets of ~{n // buckets}:"
    for form in (2, 1):
        ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, form)
        acc = {}
        for r in range(6):
            keys.sync(); vals.sync()
            if r >= 2:
                ctx.profile_begin()
            N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, keys.device, vals.device, n))
            if r >= 2:
                for name, v in ctx.profile_end().items():
                    a = acc.setdefault(name.split("<")[0], [0, 0.0])
                    a[0] += 1; a[1] += v[1]
        kk = keys.get_data()
        assert (kk[1:] >= kk[:-1]).all()
        total = sum(v[1] for v in acc.values()) / 4
        line += f"  form {form}: {total * 1e3:.1f} us (" + ", ".join(f"{nm.replace('sort_', '').replace('_kernel', '')} {v[1] / 4 * 1e3:.1f}" for nm, v in acc.items()) + ")"
    print(line)
    keys.dispose(); vals.dispose()
ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 0)
ctx.close()
