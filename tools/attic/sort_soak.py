#!/usr/bin/env python3
"""A soak of lbvh_sort_pairs in the two-level form's range (2^15 <= pairs < 2^21) and around it: random sizes, key distributions
chosen to stress the balanced bucket map (uniform; few distinct 12-bit prefixes; one prefix holding most pairs; Morton-like codes
below 2^30 with 0xFFFFFFFF pads; clustered; already sorted / reversed; all equal; keys that differ only in their low bits), each
sorted with the form left to the library's own choice (so the one-sort-stale hint is wrong every time the distribution changes:
the bucket kernel's slow path runs) and, every fourth case, forced either way.  Every result against the oracle.
usage: python tools/sort_soak.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from unitysimpleraytracing_amd import _native as N
from unitysimpleraytracing_amd.host import Context, DataBuffer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
KINDS = ["uniform", "few_prefixes", "heavy_prefix", "morton_pads", "clustered", "sorted", "reversed", "all_equal", "low_bits", "two_values", "top_bin"]


def make(kind, n):
    if kind == "uniform":
        k = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    elif kind == "few_prefixes":
        p = rng.integers(0, 4096, int(rng.integers(1, 40)), dtype=np.uint64)
        k = (rng.choice(p, n) << np.uint64(20)) | rng.integers(0, 1 << 20, n, dtype=np.uint64)
    elif kind == "heavy_prefix":
        k = rng.integers(0, 1 << 32, n, dtype=np.uint64)
        heavy = rng.random(n) < rng.uniform(0.05, 0.9)
        k[heavy] = (np.uint64(int(rng.integers(0, 4096))) << np.uint64(20)) | rng.integers(0, 1 << 20, int(heavy.sum()), dtype=np.uint64)
    elif kind == "morton_pads":
        k = rng.integers(0, 1 << 30, n, dtype=np.uint64)
        k[n - int(rng.integers(0, n // 3 + 1)):] = 0xFFFFFFFF
    elif kind == "clustered":
        k = np.minimum((rng.random(n) ** int(rng.integers(2, 6)) * (1 << 32)).astype(np.uint64), (1 << 32) - 1)
    elif kind == "sorted":
        k = np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint64))
    elif kind == "reversed":
        k = np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint64))[::-1].copy()
    elif kind == "all_equal":
        k = np.full(n, int(rng.integers(0, 1 << 32)), dtype=np.uint64)
    elif kind == "low_bits":
        k = np.uint64(int(rng.integers(0, 1 << 12)) << 20) | rng.integers(0, 1 << int(rng.integers(1, 20)), n, dtype=np.uint64)
    elif kind == "two_values":
        k = rng.choice(np.array([int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))], dtype=np.uint64), n)
    else:                                   # everything in the last fine bin, spread over its 20 bits
        k = np.uint64(0xFFF00000) | rng.integers(0, 1 << 20, n, dtype=np.uint64)
    return k.astype(np.uint32)


cases = 0
t_end = time.time() + budget
with Context(0) as ctx:
    while time.time() < t_end:
        cases += 1
        n = int(rng.choice([rng.integers(1 << 15, 1 << 21), rng.integers(1 << 15, 1 << 17), rng.integers(1 << 20, 1 << 21),
                            rng.integers(1 << 14, 1 << 15) + (1 << 14), rng.integers((1 << 21) - 5000, (1 << 21) + 5000)]))
        kind = str(rng.choice(KINDS))
        keys = make(kind, n)
        vals = rng.permutation(n).astype(np.uint32)
        form = int(rng.integers(1, 3)) if cases % 4 == 0 else 0
        ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, form)
        kb, vb = DataBuffer(ctx, n, np.uint32), DataBuffer(ctx, n, np.uint32)
        kb.local[:] = keys; vb.local[:] = vals; kb.sync(); vb.sync()
        N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, kb.device, vb.device, n))
        ok, ov = O.sort_pairs(keys, vals)
        assert (kb.get_data() == ok).all() and (vb.get_data() == ov).all(), (cases, kind, n, form)
        kb.dispose(); vb.dispose()
        if cases % 200 == 0:
            print("case", cases, kind, n, "form", form, "ok", flush=True)
    ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 0)
print("cases", cases, "all equal (seed", seed, ")")
