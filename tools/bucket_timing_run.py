import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from unitysimpleraytracing_amd import _native as N
from unitysimpleraytracing_amd.host import Context, DataBuffer
ctx = Context(0)
n, buckets = 1 << 20, 85
rng = np.random.default_rng(3)
k = rng.integers(0, 1 << 24, size=n, dtype=np.uint64) | (rng.integers(0, buckets, size=n, dtype=np.uint64) << 24)
keys = DataBuffer(ctx, n, np.uint32); vals = DataBuffer(ctx, n, np.uint32)
keys.local[:] = k.astype(np.uint32); vals.local[:] = np.arange(n, dtype=np.uint32)
ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 2)
for r in range(3):
    keys.sync(); vals.sync()
    N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, keys.device, vals.device, n))
ctx.sync()
