"""Phase timestamps of the two-level sort's bucket kernel (round 5): run against a measurement build —
    bash tools/build_variant.sh timing -DLBVH_BUCKET_TIMING
    LBVH_LIB=build_exp/liblbvh_timing.so python tools/bucket_timing_run.py
The library then prints, after every two-level sort, the s_memtime ticks of its LARGEST bucket's phase borders ([1] start of the
bucket known, [2] pairs loaded, [3 + 4p] pass p ranked, [4 + 4p] pairs exchanged, [5 + 4p] read back, [30] stored; [32] / [33] /
[34]: the last pass's cells cleared / items ranked / digits laid out).  Input: 2^20 random keys whose top byte takes 85 values
(buckets of ~12 300 pairs, the size of cfg2's largest)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from unitysimpleraytracing_amd import _native as N  # noqa: E402
from unitysimpleraytracing_amd.host import Context, DataBuffer  # noqa: E402

ctx = Context(0)
n, buckets = 1 << 20, 85
rng = np.random.default_rng(3)
k = rng.integers(0, 1 << 24, size=n, dtype=np.uint64) | (rng.integers(0, buckets, size=n, dtype=np.uint64) << 24)
keys = DataBuffer(ctx, n, np.uint32)
vals = DataBuffer(ctx, n, np.uint32)
keys.local[:] = k.astype(np.uint32)
vals.local[:] = np.arange(n, dtype=np.uint32)
ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 2)
for r in range(3):
    keys.sync(); vals.sync()
    N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, keys.device, vals.device, n))
ctx.sync()
ctx.close()
