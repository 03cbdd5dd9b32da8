#!/usr/bin/env python3
"""Randomised parity soak (GPU box): for a time budget, draw scenes of random size and shape (uniform soup, clustered
soup with duplicated triangles, degenerate slivers, tiled tori, all triangles in one Morton cell, triangles outside the
scene box), build them through lbvh_build_scene AND through the staged calls, and compare every array with the CPU
oracle; trace a random camera (inside / outside the scene, random resolution incl. ragged tiles, random shard count,
two or three frames so cost-ordered / cooperative / reprojected dispatch all run) in fast and reference mode against the
oracle's frames: reference mode against the reference's loop, the fast modes against the same loop under their accept rule (a t in
front of its own triangle's box does not count) — and the two oracle frames against each other: they may differ only
where the reference's winner is such a t (counted and listed).  Also sorts random (key, value) arrays of random size and digit structure, and every eighth case
animates and path-traces a small dynamic scene (1 .. 4 bounces) against the extension's own oracle.  Prints one line per case and a
summary; exits non-zero on the first mismatch.   usage: python tools/fuzz_parity.py [seconds] [seed] [first_case]   (FUZZ_PATH_EVERY=1: a path-tracer-heavy stream)
Every random draw of a case is made up front (draw_case), so `first_case` K replays the generator through cases 1 .. K-1 without
running them and starts at case K: the way back to a failure that a long soak found (its message names the case's number)."""
import ctypes as C, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) & 0xFFFFFF
first_case = int(sys.argv[3]) if len(sys.argv) > 3 else 1
path_every = int(os.environ.get("FUZZ_PATH_EVERY", "8"))       # 1: every case also path-traces a dynamic scene (a cfg5-heavy soak; another case stream)
rng = np.random.default_rng(seed0)
print("seed", seed0, "budget", budget, "s", "first case", first_case, flush=True)


def words(a):
    return np.ascontiguousarray(a).view(np.uint32)


artefacts = []       # (what, pixels): where the reference's record is an fp32 artefact that the fast modes' accept rule drops


def oracle_frames(b, cam, what):
    """(the reference's frame, the frame under the fast modes' accept rule).  They differ only where the reference's winner is a t in
    front of its own triangle's box — checked here on the CPU, pixel by pixel, and counted."""
    oh, _ = O.trace_primary(b, cam, threads=8)
    of, _ = O.trace_primary(b, cam, threads=8, fast_rule=True)
    if not (words(oh) == words(of)).all():
        unexplained, explained = O.unexplained_mismatches(b, cam, oh, of, words=True)
        assert not unexplained, what + ("oracle: reference against fast rule", unexplained[:4])
        artefacts.append((what, explained))
        print("   the reference's record is an fp32 artefact (a t in front of its own triangle's box) at", what, explained, flush=True)
    return oh, of


def draw_scene(kind, n):
    """The random draws of a scene (NOT the scene: a skipped case costs nothing but its draws)."""
    if kind == "torus":
        return dict(nu=int(rng.integers(6, 40)), nv=int(rng.integers(4, 24)), grid=int(rng.integers(1, 4)), seed=int(rng.integers(1 << 30)))
    return dict(seed=int(rng.integers(1 << 30)))


def make_scene(kind, n, p):
    if kind == "soup":
        return scenes.random_triangles(n, seed=p["seed"])
    if kind == "torus":
        return scenes.tiled_torus(nu=p["nu"], nv=p["nv"], grid=p["grid"], seed=p["seed"])
    t = scenes.random_triangles(n, seed=p["seed"])
    if kind == "dups":               # every triangle several times over + a cluster in one cell
        k = max(n // 4, 1)
        t[k:2 * k] = t[:k][: len(t[k:2 * k])]
        t[2 * k:3 * k] = t[:k][: len(t[2 * k:3 * k])]
        for f in ("a", "b", "c"):
            t[f][3 * k:] = t[f][3 * k:] * np.float32(1e-3)
    elif kind == "one_cell":
        for f in ("a", "b", "c"):
            t[f] = t[f] * np.float32(1e-4) + np.float32(17.0)
    elif kind == "slivers":
        t["b"] = t["a"] + (t["b"] - t["a"]) * np.float32(1e-6)
    elif kind == "outside":
        for f in ("a", "b", "c"):
            t[f] = t[f] * np.float32(3.0)
    return t


def random_camera(w, h):
    pos = tuple(float(x) for x in rng.uniform(-140, 140, 3)) if rng.random() < 0.5 else (0.0, 0.0, float(rng.uniform(150, 400)))
    cam = scenes.camera(w, h, pos)
    if rng.random() < 0.7:
        yaw, pitch = math.radians(rng.uniform(-180, 180)), math.radians(rng.uniform(-60, 60))
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        m = np.array(cam["camera_to_world"], dtype=np.float32).reshape(4, 4).copy()
        m[:3, :3] = (R @ m[:3, :3].astype(np.float64)).astype(np.float32)
        cam = dict(cam); cam["camera_to_world"] = m.reshape(-1).copy()
    return cam


def draw_case(case, skipped):
    """Every random draw of case `case`, in a fixed order (the generator's stream IS the case list: do not reorder)."""
    q = {}
    count = int(rng.choice([rng.integers(1, 5000), rng.integers(5000, 400000), rng.integers(400000, 3000000)]))
    shift = int(rng.integers(0, 25))
    keys = (rng.integers(0, 1 << 32, size=count, dtype=np.uint64) >> np.uint64(shift) << np.uint64(rng.integers(0, shift + 1)))
    pads = rng.random() < 0.3
    vals = rng.permutation(count)
    if not skipped:
        keys = keys.astype(np.uint32)
        if pads:
            keys[count - count // 5:] = 0xFFFFFFFF
        q["keys"], q["vals"] = keys, vals.astype(np.uint32)
    q["count"], q["shift"] = count, shift
    q["kind"] = str(rng.choice(["soup", "torus", "dups", "one_cell", "slivers", "outside"]))
    q["n"] = int(rng.choice([rng.integers(2, 70), rng.integers(70, 3000), rng.integers(3000, 120000)]))
    q["scene"] = draw_scene(q["kind"], q["n"])
    w, h = int(rng.integers(1, 260)), int(rng.integers(1, 140))
    q["w"], q["h"] = w, h
    q["cam"] = random_camera(w, h)
    q["shards"] = int(rng.choice([1, 1, 2, 3, 8]))
    if w >= 3 and h >= 3:
        x0, y0 = int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))
        x1, y1 = int(rng.integers(x0 + 1, w + 1)), int(rng.integers(y0 + 1, h + 1))
        q["rect"] = (x0, y0, x1, y1)
    q["cam2"] = random_camera(w, h)
    if case % path_every == 0:
        nu, nv, g = int(rng.integers(6, 28)), int(rng.integers(4, 18)), int(rng.integers(1, 4))
        sseed = int(rng.integers(1 << 30))
        sd, bounces, angle = int(rng.integers(1 << 20)), int(rng.integers(1, 5)), float(rng.uniform(0.0, 0.3))
        pw, ph = int(rng.integers(1, 140)), int(rng.integers(1, 90))
        q["path"] = dict(nu=nu, nv=nv, grid=g, seed=sseed, sd=sd, bounces=bounces, angle=angle, pw=pw, ph=ph, z=float(rng.uniform(60, 200)))
    return q


def run_case(ctx, case, q):
    # ---- a sort ------------------------------------------------------------------------------------------------
    count, shift, keys, vals = q["count"], q["shift"], q["keys"], q["vals"]
    kb = DataBuffer(ctx, count, np.uint32); vb = DataBuffer(ctx, count, np.uint32)
    kb.local[:] = keys; vb.local[:] = vals; kb.sync(); vb.sync()
    N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, kb.device, vb.device, count))
    ok, ov = O.sort_pairs(keys, vals)
    assert (kb.get_data() == ok).all() and (vb.get_data() == ov).all(), (case, "sort", count, shift)
    kb.dispose(); vb.dispose()
    # ---- a scene -----------------------------------------------------------------------------------------------
    kind, w, h, cam, shards = q["kind"], q["w"], q["h"], q["cam"], q["shards"]
    tris = make_scene(kind, q["n"], q["scene"])
    n = len(tris)
    d = RaytracingMeshDrawer(ctx, tris).awake(fast=True)
    c = d.container
    b = O.Built(tris, capacity=c.capacity, threads=8)
    for how in ("awake", "rebuild", "rebuild"):                      # staged chain, then lbvh_build_scene (plain, then graph)
        if how == "rebuild":
            c.bvh_internal_node.fill_u32(0, mirror=False); c.bvh_data.fill_u32(0x7FC00000, mirror=False)
            d.rebuild(fast=True)
        bad_leaf, bad_inner = c.get_all_gpu_data()
        assert len(bad_leaf) == 0 and len(bad_inner) == 0, (case, kind, n, how)
        assert (c.keys.local == b.keys).all() and (c.triangle_index.local == b.indices).all(), (case, kind, n, how, "keys")
        assert (words(c.bvh_internal_node.local)[: 6 * (n - 1)] == words(b.internal)[: 6 * (n - 1)]).all(), (case, kind, n, how, "internal")
        assert (words(c.bvh_leaf_node.local)[: 2 * n] == words(b.leaf)[: 2 * n]).all(), (case, kind, n, how, "leaf")
        assert (c.bvh_data.local["min"][: n - 1] == b.bvh["min"][: n - 1]).all() and (c.bvh_data.local["max"][: n - 1] == b.bvh["max"][: n - 1]).all(), (case, kind, n, how, "boxes")
    oh, of = oracle_frames(b, cam, (case, kind, n, w, h))
    for frame in range(3):
        d._hits = None if frame == 0 else d._hits
        if shards == 1:
            d.update(cam, mode=L.TRACE_FAST)
            fh = d.hits()
        else:
            for r in range(shards):
                d.update_shard(cam, r, shards, mode=L.TRACE_FAST)
            fh = d.hits()
        assert (fh["t"] == of["t"]).all(), (case, kind, n, w, h, shards, frame, "fast t")
    d.update(cam, mode=L.TRACE_REFERENCE)
    rh = d.hits()
    assert (words(rh) == words(oh)).all(), (case, kind, n, w, h, "reference hits")
    # LBVH_TRACE_FAST_EXACT: every word of the oracle's records ("dups" scenes tie on most pixels), whole frame and shards
    for frame in range(2):
        if shards == 1:
            d.update(cam, mode=L.TRACE_FAST_EXACT)
        else:
            for r in range(shards):
                d.update_shard(cam, r, shards, mode=L.TRACE_FAST_EXACT)
        assert (words(d.hits()) == words(of)).all(), (case, kind, n, w, h, shards, frame, "exact mode")
    # a sub-rectangle of the frame, two frames (its own history), then the same rectangle from a turned camera
    if "rect" in q:
        x0, y0, x1, y1 = q["rect"]
        for frame in range(2):
            d.update(cam, rect=(x0, y0, x1, y1), mode=L.TRACE_FAST)
            assert (d.hits()["t"] == of["t"][y0:y1, x0:x1]).all(), (case, kind, n, w, h, (x0, y0, x1, y1), frame, "rectangle")
    # a second camera: the history of the first one is reprojected
    cam2 = q["cam2"]
    oh2, of2 = oracle_frames(b, cam2, (case, kind, n, w, h, "second camera"))
    if "rect" in q:
        d.update(cam2, rect=(x0, y0, x1, y1), mode=L.TRACE_FAST)
        assert (d.hits()["t"] == of2["t"][y0:y1, x0:x1]).all(), (case, kind, n, w, h, "rectangle, second camera")
    d.update(cam2, mode=L.TRACE_FAST)
    assert (d.hits()["t"] == of2["t"]).all(), (case, kind, n, w, h, "second camera")
    d.update(cam2, mode=L.TRACE_FAST_EXACT)
    assert (words(d.hits()) == words(of2)).all(), (case, kind, n, w, h, "second camera, exact mode")
    d.on_destroy()
    # ---- every eighth case: the dynamic scene + path tracer (cfg5 extension) against its own oracle ------------------
    if "path" in q:
        from unitysimpleraytracing_amd.host import DynamicPathTracer
        pp = q["path"]
        ptris, body, centres = scenes.tiled_torus(nu=pp["nu"], nv=pp["nv"], grid=pp["grid"], seed=pp["seed"], with_bodies=True)
        sd, bounces, angle, pw, ph = pp["sd"], pp["bounces"], pp["angle"], pp["pw"], pp["ph"]
        pt = DynamicPathTracer(ctx, ptris, body, centres, t_min=1e-3, albedo=0.7, seed=sd)
        pt.animate(angle)
        pb = O.Built(O.animate(ptris, body, centres, angle), capacity=pt.drawer.container.capacity, threads=8)
        pcam = scenes.camera(pw, ph, (0.0, 0.0, pp["z"]))
        pt.render(pcam, bounces=bounces)
        img = pt.image()
        oimg, ost = O.path_trace(pb, pcam, bounces=bounces, t_min=1e-3, albedo=0.7, seed=sd, threads=8)
        gst = pt.states.get_data()[: pw * ph]
        # every path state and pixel: both sides resolve exact t ties to the lowest triangle index
        assert (gst["origin"] == ost["origin"]).all() and (gst["dir"] == ost["dir"]).all(), (case, "path", len(ptris), pw, ph, bounces)
        assert (img.view(np.uint16) == oimg.view(np.uint16)).all(), (case, "path image", len(ptris), pw, ph, bounces)
        pt.drawer.on_destroy()
    print(f"case {case}: sort {count} >> {shift}; {kind} n={n} {w}x{h} shards {shards}: ok", flush=True)


if __name__ == "__main__":
    case = 0
    while case + 1 < first_case:                      # replay the generator up to the first case asked for
        case += 1
        draw_case(case, skipped=True)
    ran = 0
    t_end = time.time() + budget
    with Context(0) as ctx:
        while time.time() < t_end:
            case += 1
            run_case(ctx, case, draw_case(case, skipped=False))
            ran += 1
    print("cases", ran, "all equal", f"(cases {first_case} .. {case} of seed {seed0});",
          len(artefacts), "camera(s) with a pixel where the reference's record is an fp32 artefact that the fast modes' rule drops:", artefacts)
