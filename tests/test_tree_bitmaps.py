"""CPU model of the tree kernel's search-free form (csrc/lbvh_build.hip, tree_body with bitmaps) against the oracle's literal
Karras searches (oracle/lbvh_oracle.c: DetermineRange / FindSplit, BVH.compute:35-92).

For sorted UNIQUE keys clz(k_i ^ k_j) = min_{i <= m < j} delta_m with delta_m = clz(k_m ^ k_{m+1}): the radix tree is the
Cartesian tree of the adjacent-key prefix array.  Both searches of the reference are "last position where a monotone
predicate holds", so
    dmin = min(delta_{i-1}, delta_i)                       (delta_{-1} = delta_{n-1} = -1)
    other end of node i = the nearest m in direction d with delta_m <= dmin
    split               = the first m >= first with delta_m <= clz(k_first ^ k_last)
The kernel answers these with nearest-set-bit lookups in bitmaps B[v] = {m : delta_m <= v} of its 768-key LDS window (built by
a 32 x 32 bit transpose across lanes); a node whose lookup leaves the window goes to the wave-wide search that was there before.
This file models exactly that (windows, unknown positions, sentinels, the transpose network, the summary word) in numpy / Python
and compares (first, last, split) -> the node arrays with the oracle on adversarial key sets.
"""
import numpy as np
import pytest

import oracle as O
from unitysimpleraytracing_amd import layouts as L

THREADS, HALO = 256, 256
WINDOW = THREADS + 2 * HALO
WORDS = WINDOW // 32


def clz32(v):
    v = int(v)
    return 32 - v.bit_length()


def transpose32_lanes(rows):
    """the kernel's butterfly: 32 lanes, lane l holds a 32-bit row; five exchanges with lane l ^ j (ds_swizzle) leave lane v
    holding column v (bit p of lane v = bit v of the row lane p started with)"""
    x = [int(r) for r in rows]
    for s, m in enumerate((0x55555555, 0x33333333, 0x0F0F0F0F, 0x00FF00FF, 0x0000FFFF)):
        j = 1 << s
        y = [x[l ^ j] for l in range(32)]
        for l in range(32):
            if l & j:
                x[l] = (x[l] & ~m & 0xFFFFFFFF) | ((y[l] >> j) & m)
            else:
                x[l] = (x[l] & m) | ((y[l] << j) & ~m & 0xFFFFFFFF)
    return x


def test_transpose_network_is_a_transpose():
    rng = np.random.default_rng(5)
    rows = rng.integers(0, 2**32, 32, dtype=np.uint64)
    t = transpose32_lanes(rows)
    for v in range(32):
        for p in range(32):
            assert (t[v] >> p) & 1 == (int(rows[p]) >> v) & 1


def window_bitmaps(keys, n, w0, w1):
    """bits[word][v] (32 positions per word) and sum[v] (one bit per non-empty word) of the window [w0, w1)"""
    bits = np.zeros((WORDS, 32), dtype=np.uint64)
    for c in range(WORDS):
        rows = []
        for lane in range(32):
            m = w0 + c * 32 + lane
            if m + 1 < w1:
                d = clz32(int(keys[m]) ^ int(keys[m + 1]))
                rows.append((0xFFFFFFFF << d) & 0xFFFFFFFF if d < 32 else 0)
            elif m + 1 == n and m < w1:
                rows.append(0xFFFFFFFF)          # delta_{n-1} = -1: below every value
            else:
                rows.append(0)                   # unknown (the key past the window) or no such position
        bits[c, :] = transpose32_lanes(rows)
    sums = np.zeros(32, dtype=np.uint64)
    for v in range(32):
        for c in range(WORDS):
            if bits[c, v]:
                sums[v] |= np.uint64(1 << c)
    return bits, sums


def word(bits, q, v):
    return int(bits[q, v]) if 0 <= q < WORDS else 0


def first_set_at_or_after(bits, sums, v, s):
    """window position of the first set bit of B[v] at position >= s, or None"""
    q = s >> 5
    w = (word(bits, q, v) | (word(bits, q + 1, v) << 32)) >> (s & 31)
    if w:
        return s + ((w & -w).bit_length() - 1)
    rest = int(sums[v]) & ~((4 << q) - 1)
    if not rest:
        return None
    c = (rest & -rest).bit_length() - 1
    x = word(bits, c, v)
    return c * 32 + ((x & -x).bit_length() - 1)


def last_set_at_or_before(bits, sums, v, e):
    if e < 0:
        return None
    q = e >> 5
    w = ((word(bits, q, v) << 32) | word(bits, q - 1, v)) << (31 - (e & 31))
    w &= (1 << 64) - 1
    if w:
        return e - (64 - w.bit_length())
    rest = int(sums[v]) & (((1 << (q - 1)) - 1) if q >= 1 else 0)
    if not rest:
        return None
    c = rest.bit_length() - 1
    return c * 32 + word(bits, c, v).bit_length() - 1


def model_nodes(keys, n):
    """(first, last, split, wide) per internal node, the way the kernel's workgroups find them"""
    out = np.zeros((n - 1, 4), dtype=np.int64)
    for b0 in range(0, n - 1, THREADS):
        w0, w1 = max(b0 - HALO, 0), min(b0 + THREADS + HALO, n)
        bits, sums = window_bitmaps(keys, n, w0, w1)
        for i in range(b0, min(b0 + THREADS, n - 1)):
            me = int(keys[i])
            dl = clz32(me ^ int(keys[i - 1])) if i > 0 else -1
            dr = clz32(me ^ int(keys[i + 1])) if i + 1 < n else -1
            d = (dr > dl) - (dr < dl)
            assert d != 0
            dmin = min(dl, dr)
            wide, j = False, None
            if d > 0:
                if dmin < 0:
                    if w1 == n:
                        j = n - 1
                    else:
                        wide = True
                else:
                    p = first_set_at_or_after(bits, sums, dmin, i - w0)
                    if p is None:
                        wide = True
                    else:
                        j = w0 + p
            else:
                p = last_set_at_or_before(bits, sums, dmin, i - 1 - w0)
                if p is not None:
                    j = w0 + p + 1
                elif w0 == 0:
                    j = 0
                else:
                    wide = True
            if wide:
                out[i] = (0, 0, 0, 1)
                continue
            first, last = min(i, j), max(i, j)
            assert w0 <= first and last < w1
            node_delta = clz32(int(keys[first]) ^ int(keys[last]))
            p = first_set_at_or_after(bits, sums, node_delta, first - w0)
            assert p is not None and w0 + p < last
            out[i] = (first, last, w0 + p, 0)
    return out


def oracle_nodes(keys, n):
    """(first, last, split) of every node from the oracle's node arrays: split = leftNode; the range from the parents
    (lbvh_build.hip node_range)"""
    internal, _ = O.build_tree(keys, n)
    left = internal["leftNode"].astype(np.int64)
    parent = internal["parent"].astype(np.int64)
    out = np.zeros((n - 1, 3), dtype=np.int64)
    for p in range(n - 1):
        first_kind = p <= left[p]
        other = n - 1 if first_kind else 0
        c = parent[p] if p else -1
        while 0 <= c < n - 1:
            if (c <= left[c]) != first_kind:
                other = c
                break
            if c == 0:
                break
            c = parent[c]
        out[p] = (p, other, left[p]) if first_kind else (other, p, left[p])
    return out


def key_sets():
    rng = np.random.default_rng(11)
    yield "dense", np.arange(3000, dtype=np.uint32)
    yield "dense_offset", np.arange(2100, dtype=np.uint32) + np.uint32(0x3FFFF000)
    yield "random", np.unique(rng.integers(0, 2**30, 2500, dtype=np.uint32))
    yield "powers", np.unique(np.concatenate([np.uint32(1) << np.arange(31, dtype=np.uint32), np.arange(1500, dtype=np.uint32) * 3]))
    # strictly nested prefixes: key m differs from its successor at an ever lower bit, then clusters
    nested = [0]
    for b in range(30, 0, -1):
        nested.append(nested[-1] + (1 << b) // 2 + 1)
    yield "nested", np.unique(np.concatenate([np.array(nested, dtype=np.uint32), np.arange(900, dtype=np.uint32) + np.uint32(1 << 29)]))
    # clusters of 250 .. 600 consecutive keys separated by large gaps: ranges that just fit / just leave the 768 window
    parts, base = [], 0
    for size in (250, 255, 256, 257, 300, 511, 512, 513, 600, 64, 1, 2, 767, 768, 769):
        parts.append(np.arange(size, dtype=np.uint64) + base)
        base += 1 << 22
    yield "clusters", np.concatenate(parts).astype(np.uint32)
    yield "two", np.array([5, 9], dtype=np.uint32)
    yield "three", np.array([0, 1, 0x80000000], dtype=np.uint32)
    yield "high_bit", np.unique(rng.integers(0, 2**32, 1800, dtype=np.uint64).astype(np.uint32))
    # what DistributeKeys leaves behind for a scene of many equal Morton codes: steps of 1 with rare jumps
    steps = np.where(rng.random(2600) < 0.02, rng.integers(1, 1 << 20, 2600), 1)
    yield "distributed", np.concatenate([[0], np.cumsum(steps)]).astype(np.uint32)


@pytest.mark.parametrize("name,keys", list(key_sets()), ids=[k for k, _ in key_sets()])
def test_bitmap_lookups_equal_the_karras_searches(name, keys):
    keys = np.ascontiguousarray(np.sort(keys))
    n = len(keys)
    assert len(np.unique(keys)) == n
    want = oracle_nodes(keys, n)
    got = model_nodes(keys, n)
    narrow = got[:, 3] == 0
    assert (got[narrow, :3] == want[narrow]).all()
    # a node is handed to the wave-wide search exactly when its range (or the search for it) leaves its workgroup's window
    for i in map(int, np.nonzero(~narrow)[0]):
        b0 = (i // THREADS) * THREADS
        w0, w1 = max(b0 - HALO, 0), min(b0 + THREADS + HALO, n)
        first, last, _ = map(int, want[i])
        d_right = first == i
        # the end the lookup could not see: the position of the closing delta is outside the known part of the window
        closing = last if d_right else first - 1
        known_hi = w1 - 2 if w1 < n else w1 - 1
        assert closing > known_hi or closing < w0, (name, i, first, last, w0, w1)
        # ... and once its wave has found the range in memory, the split still comes from the window's bitmaps whenever the left
        # child ends inside the window (tree_body, the wide loop): the first member of B[delta_node] at or after `first`
        if w0 <= first < w1 and keys[first] != keys[last]:
            bits, sums = window_bitmaps(keys, n, w0, w1)
            p = first_set_at_or_after(bits, sums, clz32(int(keys[first]) ^ int(keys[last])), int(first) - w0)
            if p is not None and w0 + p < last:
                assert w0 + p == want[i][2], (name, i)
    if n > WINDOW:
        assert (~narrow).any()          # the root at least


def test_nearest_smaller_value_form_on_random_small_arrays():
    """the statement itself, without windows: 300 arrays, n <= 200"""
    rng = np.random.default_rng(3)
    for trial in range(300):
        n = int(rng.integers(2, 200))
        bits = int(rng.integers(8, 32))
        keys = np.unique(rng.integers(0, 2**bits, n, dtype=np.uint64).astype(np.uint32))
        n = len(keys)
        if n < 2:
            continue
        want = oracle_nodes(keys, n)
        delta = [-1] + [clz32(int(keys[m]) ^ int(keys[m + 1])) for m in range(n - 1)] + [-1]      # delta[m + 1] = delta_m
        for i in range(n - 1):
            dl, dr = delta[i], delta[i + 1]
            dmin = min(dl, dr)
            if dr > dl:
                j = next(m for m in range(i, n) if delta[m + 1] <= dmin)
            else:
                j = next(m for m in range(i - 1, -2, -1) if delta[m + 1] <= dmin) + 1
            first, last = min(i, j), max(i, j)
            nd = clz32(int(keys[first]) ^ int(keys[last]))
            split = next(m for m in range(first, last) if delta[m + 1] <= nd)
            assert (first, last, split) == tuple(want[i]), (trial, i)
