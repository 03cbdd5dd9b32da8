"""The two-level sort's bucket map (lbvh_sort.hip: build_bucket_map) restated in numpy: bucket(bin) = floor(pairs before the bin x
floor(2^40 / count) / 2^32), clamped to 255.  Whatever the histogram, the map must be monotone (buckets in key order = the stable
partition is a step of a sort), stay below 256, and leave no bucket larger than count / 256 + its largest bin + 1; the pass count of
a bucket is the number of bytes of its key span.  (The kernels' own results are checked on the GPU: every sort test and the soak.)"""
import numpy as np
import pytest

FINE_BINS = 4096


def bucket_map(fine, count):
    mul = (1 << 40) // count
    before = np.concatenate([[0], np.cumsum(fine)[:-1]]).astype(np.uint64)
    return np.minimum((before * np.uint64(mul)) >> np.uint64(32), 255).astype(np.int64)


def passes_of(first_bin, last_bin, fine_shift):
    base = first_bin << fine_shift
    top = 0xFFFFFFFF if last_bin >= FINE_BINS - 1 else ((last_bin + 1) << fine_shift) - 1
    span = top - base
    return 1 if span == 0 else (span.bit_length() + 7) // 8


@pytest.mark.parametrize("kind", ["uniform", "morton_like", "one_bin", "two_heavy_bins", "sparse", "pads"])
@pytest.mark.parametrize("count", [1 << 15, 300001, (1 << 21) - 1])
def test_bucket_map_is_monotone_bounded_and_balanced(kind, count):
    rng = np.random.default_rng(7)
    if kind == "uniform":
        bins = rng.integers(0, FINE_BINS, count)
    elif kind == "morton_like":
        bins = np.minimum((rng.random(count) ** 2 * 1024).astype(np.int64), 1023)
    elif kind == "one_bin":
        bins = np.full(count, 1234)
    elif kind == "two_heavy_bins":
        bins = np.where(rng.random(count) < 0.5, 7, 4000)
    elif kind == "sparse":
        bins = rng.choice([3, 500, 501, 2900, 4095], count)
    else:
        bins = rng.integers(0, 1024, count)
        bins[count - count // 5:] = FINE_BINS - 1
    fine = np.bincount(bins, minlength=FINE_BINS)
    m = bucket_map(fine, count)
    assert (np.diff(m) >= 0).all() and m.min() >= 0 and m.max() <= 255
    sizes = np.bincount(m, weights=fine, minlength=256)
    assert sizes.sum() == count
    assert sizes.max() <= count // 256 + fine.max() + 1
    # every bucket's keys lie between its first and last non-empty bin, and the spans are disjoint and ordered
    last_seen = -1
    for b in range(256):
        own = np.nonzero((m == b) & (fine > 0))[0]
        if len(own) == 0:
            continue
        assert own[0] > last_seen
        last_seen = own[-1]


def chosen_map(fine, count, fine_shift):
    """build_bucket_map's choice: the balanced map or the plain one (bin >> 4), whichever has the smaller largest pairs x passes"""
    def cost(m):
        worst = 0
        for b in range(256):
            own = np.nonzero((m == b) & (fine > 0))[0]
            if len(own):
                worst = max(worst, int(fine[own].sum()) * passes_of(int(own[0]), int(own[-1]), fine_shift))
        return worst
    bal, plain = bucket_map(fine, count), np.arange(FINE_BINS) >> 4
    # (the kernel prices a plain bucket by its full 16-bin span; a balanced one by its non-empty bins)
    plain_cost = max(int(fine[16 * b: 16 * b + 16].sum()) * passes_of(16 * b, 16 * b + 15, fine_shift) for b in range(256))
    return ("plain", plain) if plain_cost <= cost(bal) else ("balanced", bal)


def test_evenly_spread_keys_keep_the_aligned_top_byte_and_clustered_ones_get_balanced_ranges():
    rng = np.random.default_rng(3)
    count = 1 << 20
    uniform = np.bincount(rng.integers(0, FINE_BINS, count), minlength=FINE_BINS)
    assert chosen_map(uniform, count, 20)[0] == "plain"          # 17 bins of 32-bit keys would need a fourth pass
    clustered = np.bincount(np.minimum((rng.random(count) ** 3 * FINE_BINS).astype(np.int64), FINE_BINS - 1), minlength=FINE_BINS)
    assert chosen_map(clustered, count, 18)[0] == "balanced"


def test_pass_counts_follow_the_span():
    assert passes_of(0, 0, 18) == 3               # one bin of Morton codes: 18 bits
    assert passes_of(10, 50, 18) == 3             # 41 bins x 2^18 < 2^24
    assert passes_of(10, 74, 18) == 4             # 65 bins: 25 bits
    assert passes_of(4095, 4095, 18) == 4         # the last bin holds everything up to 0xFFFFFFFF
    assert passes_of(4095, 4095, 20) == 3         # ... which is 20 bits wide when the bins are the top 12 bits
    assert passes_of(0, 4095, 20) == 4
