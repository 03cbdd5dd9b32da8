"""Where the fast modes' records differ from the reference's, and why they no longer depend on the visiting order,
pinned by the case a 3000-second fuzz soak found (tests/golden/grazing_ray_case.json: tools/fuzz_parity.py, seed 101, case 19899).

The reference's traversal prunes nothing (Raytracing.compute:133-176): its record for a pixel is the minimum computed t over EVERY
triangle whose padded box the ray's line passes.  The triangle test (:37-73) is ill-conditioned for a ray almost inside a
triangle's plane: here det = 6.07e-4, both dot products cancel to exactly 0, u = v = 0.0 and t = 109.23336 — a "hit" 0.094 in
front of the distance at which the ray enters that triangle's own box (109.32758); in double precision the ray misses the triangle
(v = -0.044).  A walk that skips boxes entered beyond its best hit so far saw that record or not depending on the order in which it
met the leaves (the 8x8 packet found the genuine hit 109.25598 first and never opened the box; the same ray walked alone opened it
first).  Since round 5 the fast modes therefore do not count a t in front of its own triangle's box (lbvh_rt.h: hit_counts); both
CPU restatements carry the same rule as an option (`fast_rule`), and the fast modes equal THAT frame exactly, whatever the order.
The reference's frame and the fast-rule frame differ exactly where the reference's winner is such a t."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O                                                   # noqa: E402
from oracle import literal_emulation as E                            # noqa: E402
from unitysimpleraytracing_amd import layouts as L, scenes           # noqa: E402

CASE = json.load(open(os.path.join(ROOT, "tests", "golden", "grazing_ray_case.json")))


def case_inputs():
    p = CASE["scene"]
    tris = scenes.tiled_torus(nu=p["nu"], nv=p["nv"], grid=p["grid"], seed=p["seed"])
    cam = dict(CASE["camera"])
    cam["camera_to_world"] = np.array(cam["camera_to_world"], dtype=np.float32)
    return tris, cam


@pytest.fixture(scope="module")
def built():
    tris, cam = case_inputs()
    return tris, cam, O.Built(tris, threads=O.num_threads())


def test_the_c_oracle_reports_the_artefact_and_says_it_lies_before_its_box(built):
    tris, cam, b = built
    x, y = CASE["pixel"]
    oh, _ = O.trace_primary(b, cam, rect=(x, y, x + 1, y + 1))
    r, want = oh[0, 0], CASE["reference_record"]
    assert (float(r["t"]), int(r["tri"]), float(r["u"]), float(r["v"])) == (np.float32(want["t"]), want["tri"], 0.0, 0.0)
    assert O.winner_before_its_box(b, cam, x, y, r)
    o, d, inv = O.make_ray(cam, x, y)
    box = b.triangle_aabb[want["tri"]]
    entry = O.box_entry(box["min"], box["max"], o, inv)
    assert entry == np.float32(CASE["box_entry"]) and float(r["t"]) < float(entry)
    # the genuine hit behind it: what a pruning walker that met it first is left with
    other = CASE["pruned_walk_record"]
    assert O.ray_triangle(o, d, tris[other["tri"]]) == np.float32(other["t"])
    assert float(r["t"]) < other["t"] < float(entry)
    # under the fast modes' accept rule the oracle reports that genuine hit, every word of it
    of, _ = O.trace_primary(b, cam, rect=(x, y, x + 1, y + 1), fast_rule=True)
    assert (float(of[0, 0]["t"]), int(of[0, 0]["tri"])) == (float(np.float32(other["t"])), other["tri"])
    # double precision: the ray misses the winning triangle (v < 0) — the record is noise of the fp32 test, not geometry
    a, bb, c = (tris[want["tri"]][k].astype(np.float64) for k in ("a", "b", "c"))
    o64, d64 = np.asarray(o, dtype=np.float64), np.asarray(d, dtype=np.float64)
    e1, e2 = bb - a, c - a
    pv = np.cross(d64, e2)
    det = e1 @ pv
    v64 = (d64 @ np.cross(o64 - a, e1)) / det
    assert abs(det) < 1e-3 and v64 < -0.01


def test_the_thread_per_id_emulation_of_the_reference_kernel_reports_the_same_record(built):
    """the independent restatement (oracle/literal_emulation.py: the Raytracing kernel's thread for that pixel over arrays built by
    the emulated Awake()) agrees with the C oracle on this pixel and on its neighbours"""
    tris, cam, b = built
    x, y = CASE["pixel"]
    s = E.Scene(b.indices[: b.n], b.triangle_aabb["min"], b.triangle_aabb["max"],
                np.ascontiguousarray(b.internal).view(np.uint32).reshape(-1, 6), np.ascontiguousarray(b.leaf).view(np.uint32).reshape(-1, 2),
                b.bvh["min"], b.bvh["max"], tris["a"], tris["b"], tris["c"])
    oh, _ = O.trace_primary(b, cam, rect=(x - 1, y - 1, x + 2, y + 2))
    of, _ = O.trace_primary(b, cam, rect=(x - 1, y - 1, x + 2, y + 2), fast_rule=True)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            t, tri, u, v = E.raytracing_thread(s, cam, x + dx, y + dy, [0, 0, 0, 0])
            r = oh[dy + 1, dx + 1]
            assert (np.float32(t), int(tri), np.float32(u), np.float32(v)) == (r["t"], int(r["tri"]), r["u"], r["v"]), (dx, dy)
            t, tri, u, v = E.raytracing_thread(s, cam, x + dx, y + dy, [0, 0, 0, 0], fast_rule=True)
            r = of[dy + 1, dx + 1]
            assert (np.float32(t), int(tri), np.float32(u), np.float32(v)) == (r["t"], int(r["tri"]), r["u"], r["v"]), (dx, dy, "fast rule")


def test_the_two_semantics_differ_at_that_pixel_only(built):
    tris, cam, b = built
    oh, _ = O.trace_primary(b, cam, threads=O.num_threads())
    of, _ = O.trace_primary(b, cam, threads=O.num_threads(), fast_rule=True)
    ys, xs = np.nonzero(oh["t"] < L.MAX_FLOAT)
    found = [(int(x), int(y)) for y, x in zip(ys, xs) if O.winner_before_its_box(b, cam, x, y, oh[y, x])]
    assert found == [tuple(CASE["pixel"])]
    unexplained, explained = O.unexplained_mismatches(b, cam, oh, of, words=True)
    assert unexplained == [] and explained == [(CASE["pixel"][1], CASE["pixel"][0])]


@pytest.mark.gpu
def test_fast_modes_equal_the_fast_rule_frame_in_any_order_and_reference_mode_the_reference(built):
    from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer
    tris, cam, b = built
    oh, _ = O.trace_primary(b, cam, threads=O.num_threads())
    of, _ = O.trace_primary(b, cam, threads=O.num_threads(), fast_rule=True)
    x, y = CASE["pixel"]

    def w32(a):
        return np.ascontiguousarray(a).view(np.uint32)
    with Context(0) as ctx:
        d = RaytracingMeshDrawer(ctx, tris).awake(fast=True)
        d.update(cam, mode=L.TRACE_REFERENCE)
        assert (w32(d.hits()) == w32(oh)).all()
        for mode in (L.TRACE_FAST, L.TRACE_FAST_EXACT):
            for shards in (1, 3, 8):
                for frame in range(2):              # without and with dispatch history
                    if shards == 1:
                        d.update(cam, mode=mode)
                    else:
                        for r in range(shards):
                            d.update_shard(cam, r, shards, mode=mode)
                    fh = d.hits()
                    assert (fh["t"] == of["t"]).all(), (mode, shards, frame)
                    if mode == L.TRACE_FAST_EXACT:
                        assert (w32(fh) == w32(of)).all(), (shards, frame)
            # the visiting order no longer matters: the ray alone, its row, its tile and the frame agree on the pixel
            for rect in ((x, y, x + 1, y + 1), (0, y, CASE["w"], y + 1), (208, 8, 216, 16), (x - 3, y - 5, x + 9, y + 2)):
                d.update(cam, rect=rect, mode=mode)
                assert w32(d.hits()[y - rect[1], x - rect[0]]).tolist() == w32(of[y, x]).tolist(), (mode, rect)
        d.on_destroy()


def test_entry_distances_grow_from_a_box_to_any_box_inside_it():
    """The step of DESIGN 2.4's argument that carries the weight: with the slab test's own fp32 arithmetic ((plane - origin) * inv,
    min / max) the entry distance of a box is never larger than that of a box inside it — subtraction and multiplication by one
    factor are monotone under rounding — so `t >= entry(leaf)` implies `t >= entry(every ancestor)`.  200 000 random rays (any
    signs, tiny and huge components, origins inside and outside) against random nested boxes; infinite inverse directions included
    wherever they produce no NaN (0 * inf: there both restatements fall back to the non-NaN operand, and the library orders no
    planes by sign)."""
    rng = np.random.default_rng(11)
    f = np.float32
    n = 200000
    lo_in = rng.uniform(-100, 100, (n, 3)).astype(f)
    hi_in = lo_in + rng.uniform(0.001, 20, (n, 3)).astype(f)
    lo_out = lo_in - rng.uniform(0, 30, (n, 3)).astype(f) * (rng.random((n, 3)) < 0.7)
    hi_out = hi_in + rng.uniform(0, 30, (n, 3)).astype(f) * (rng.random((n, 3)) < 0.7)
    lo_out, hi_out = lo_out.astype(f), hi_out.astype(f)
    o = rng.uniform(-150, 150, (n, 3)).astype(f)
    d = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-6, 0, (n, 3))
    d[rng.random((n, 3)) < 0.02] = 0.0                         # exact zeros: inverse = inf
    d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)).astype(f)
    with np.errstate(all="ignore"):
        inv = (f(1) / d).astype(f)

        def entry(lo, hi):
            t1 = ((lo - o) * inv).astype(f)
            t2 = ((hi - o) * inv).astype(f)
            return np.fmax(np.fmax(np.fmin(t1[:, 0], t2[:, 0]), np.fmin(t1[:, 1], t2[:, 1])), np.fmin(t1[:, 2], t2[:, 2])), t1, t2
        e_in, a1, a2 = entry(lo_in, hi_in)
        e_out, b1, b2 = entry(lo_out, hi_out)
    clean = ~(np.isnan(a1).any(axis=1) | np.isnan(a2).any(axis=1) | np.isnan(b1).any(axis=1) | np.isnan(b2).any(axis=1))
    assert clean.sum() > 0.9 * n
    assert (e_out[clean] <= e_in[clean]).all()
    # and the oracle's scalar helper is that arithmetic
    for k in np.nonzero(clean)[0][:200]:
        assert O.box_entry(lo_in[k], hi_in[k], o[k], inv[k]) == e_in[k]
