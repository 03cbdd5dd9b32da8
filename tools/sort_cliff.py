#!/usr/bin/env python3
"""what the two-level sort's slow path costs.  lbvh_sort_pairs on adversarial inputs IMMEDIATELY AFTER a
streak of balanced sorts (the context's hint says "two-level"), against the same input under the forced four-pass form:
   equal     2^21 - 1 pairs with ONE key
   prefix    2^21 - 1 pairs whose top 12 bits are equal, the low 20 random
   half      half of the pairs in one 12-bit prefix (random low bits), the rest spread
   buckets64 64 prefixes of 16 400 pairs each (the round-5 figure)
Results are compared with the oracle.   usage: python tools/sort_cliff.py   (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from unitysimpleraytracing_amd import _native as N
from unitysimpleraytracing_amd.host import Context, DataBuffer

n = (1 << 21) - 1
rng = np.random.default_rng(7)
spread = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
vals = rng.permutation(n).astype(np.uint32)
cases = {
    "equal": np.full(n, 0x12345678, dtype=np.uint32),
    "prefix": (np.uint32(0xABC) << np.uint32(20)) | rng.integers(0, 1 << 20, n, dtype=np.uint32),
    "half": np.where(rng.random(n) < 0.5, (np.uint32(0x123) << np.uint32(20)) | rng.integers(0, 1 << 20, n, dtype=np.uint32), spread).astype(np.uint32),
    "buckets64": None,
}
b64 = np.concatenate([(np.uint32(p * 64 + 5) << np.uint32(20)) | rng.integers(0, 1 << 20, 16400, dtype=np.uint32) for p in range(64)])
cases["buckets64"] = np.concatenate([b64, rng.integers(0, 1 << 32, n - len(b64), dtype=np.uint64).astype(np.uint32)])


def timed_sort(ctx, keys, check=True):
    kb, vb = DataBuffer(ctx, n, np.uint32), DataBuffer(ctx, n, np.uint32)
    kb.local[:] = keys; vb.local[:] = vals; kb.sync(); vb.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.profile_begin()
    ctx.record(e0)
    N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, kb.device, vb.device, n))
    ctx.record(e1)
    prof = ctx.profile_end()
    ms = sum(v[1] for v in prof.values())
    form = "two-level" if any("sort_bucket_kernel" in k for k in prof) else "four-pass"
    if check:
        ok, ov = O.sort_pairs(keys, vals)
        assert (kb.get_data() == ok).all() and (vb.get_data() == ov).all()
    kb.dispose(); vb.dispose()
    return ms, form


with Context(0) as ctx:
    print(f"{n} pairs; times = sum of the sort's kernels (library events)")
    for name, keys in cases.items():
        for _ in range(5):
            ms_s, form_s = timed_sort(ctx, spread, check=False)      # the streak: the hint says two-level
        ms_hit, form_hit = timed_sort(ctx, keys)                     # the adversarial input right behind it
        ms_next, form_next = timed_sort(ctx, keys)                   # the same input again: the hint has seen it
        ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 1)
        ms_four, _ = timed_sort(ctx, keys)
        ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 0)
        print(f"{name:10s} spread input {ms_s * 1e3:8.1f} us ({form_s});  first hit {ms_hit * 1e3:9.1f} us ({form_hit});  "
              f"second {ms_next * 1e3:8.1f} us ({form_next});  forced four-pass {ms_four * 1e3:8.1f} us;  first hit / four-pass {ms_hit / ms_four:6.1f} x")
