"""Pins the CPU oracle: hand-derived known-answer vectors, the committed fixtures of the
independent thread-per-id emulation of the reference kernels (oracle/literal_emulation.py -> tests/golden/*.npz),
the literal emulation of the reference's five sort kernels, the reference's own runtime invariants, and its
slab-test debug fixture.  CPU only."""
import os

import numpy as np
import pytest

import oracle as O
from unitysimpleraytracing_amd import layouts as L
from unitysimpleraytracing_amd import scenes

F = 0xFFFFFFFF
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rows(a):
    return [tuple(int(x) for x in r) for r in a]


# ---- a-6 + a-7 known answers ---------------------------------------------------------------------

def test_kat_unique_keys():
    k = O.distribute_keys(np.array([1, 2, 4, 5, 19, 24, 25, 30], dtype=np.uint32), 8)
    assert k.tolist() == [0, 1, 3, 4, 18, 23, 24, 29]
    inner, leaf = O.build_tree(k, 8)
    # (leftNode, leftNodeType, rightNode, rightNodeType, parent, index); type 0 = internal, 1 = leaf
    assert rows(inner[:7]) == [(3, 0, 4, 0, F, 0), (0, 1, 1, 1, 2, 1), (1, 0, 2, 1, 3, 2), (2, 0, 3, 1, 0, 3),
                               (5, 0, 6, 0, 0, 4), (4, 1, 5, 1, 4, 5), (6, 1, 7, 1, 4, 6)]
    assert rows(leaf) == [(1, 0), (1, 1), (2, 2), (3, 3), (5, 4), (5, 5), (6, 6), (6, 7)]


def test_kat_duplicate_keys():
    k = O.distribute_keys(np.array([5, 5, 5, 9, 9, 12], dtype=np.uint32), 6)
    assert k.tolist() == [0, 1, 2, 6, 7, 10]
    inner, leaf = O.build_tree(k, 6)
    assert rows(inner[:5]) == [(4, 0, 5, 1, F, 0), (0, 1, 1, 1, 2, 1), (1, 0, 2, 1, 4, 2), (3, 1, 4, 1, 4, 3),
                               (2, 0, 3, 0, 0, 4)]
    assert rows(leaf) == [(1, 0), (1, 1), (2, 2), (3, 3), (3, 4), (0, 5)]


def test_distribute_is_prefix_sum_of_max_diff_1():
    rng = np.random.default_rng(5)
    keys = np.sort(rng.integers(0, 1 << 30, size=5000, dtype=np.uint32) >> np.uint64(rng.integers(0, 20)))
    pads = np.full(120, F, dtype=np.uint32)
    out = O.distribute_keys(np.concatenate([keys, pads]), 5000)
    diff = np.maximum(np.diff(keys.astype(np.int64)), 1)
    assert out[:5000].tolist() == np.concatenate([[0], np.cumsum(diff)]).tolist()
    assert (out[5000:] == F).all()                      # pads untouched
    assert (np.diff(out[:5000].astype(np.int64)) > 0).all()


def test_tree_rejects_n_below_2():
    with pytest.raises(ValueError):
        O.build_tree(np.array([0], dtype=np.uint32), 1)


def test_two_leaves():
    inner, leaf = O.build_tree(np.array([0, 7], dtype=np.uint32), 2)
    assert rows(inner[:1]) == [(0, 1, 1, 1, F, 0)]
    assert rows(leaf) == [(0, 0), (0, 1)]


# ---- a-1 ---------------------------------------------------------------------------------------------

def test_morton_known_values():
    # one triangle whose padded-AABB centre sits exactly on a lattice point
    t = np.zeros(4, dtype=L.TRIANGLE)
    t["a"][0] = (-125.0, -125.0, -125.0); t["b"][0] = t["a"][0]; t["c"][0] = t["a"][0]      # -> code 0
    t["a"][1] = (125.0, 125.0, 125.0); t["b"][1] = t["a"][1]; t["c"][1] = t["a"][1]         # -> clamp 1023^3
    t["a"][2] = (0.0, -125.0, -125.0); t["b"][2] = t["a"][2]; t["c"][2] = t["a"][2]         # x = 512
    t["a"][3] = (-125.0, -125.0, 0.0); t["b"][3] = t["a"][3]; t["c"][3] = t["a"][3]         # z = 512
    keys, idx, aabb = O.morton_aabb(t, capacity=6)
    assert keys[0] == 0
    assert keys[1] == 0x3FFFFFFF
    assert keys[2] == (1 << 29)            # bit 9 of x lands on bit 3*9+2
    assert keys[3] == (1 << 27)            # bit 9 of z lands on bit 3*9
    assert idx[:4].tolist() == [0, 1, 2, 3]
    assert keys[4:].tolist() == [F, F] and idx[4:].tolist() == [F, F]
    assert np.allclose(aabb["min"][2], (-0.001, -125.001, -125.001)) and aabb["_dummy0"][2] == 0
    assert np.allclose(aabb["max"][2], (0.001, -124.999, -124.999)) and aabb["_dummy1"][2] == 0


def test_reference_scene_statistics():
    """80x80 grid + the camera of Scene.unity: the numbers SURVEY.md section 4 derived by
    independent emulation."""
    g = scenes.grid_scene()
    assert len(g) == 12800
    keys, _, _ = O.morton_aabb(g)
    assert len(np.unique(keys)) == 1156
    b = O.Built(g, capacity=scenes.capacity_for(len(g)))
    assert np.allclose(b.bvh["min"][0], (-4.001, -4.001, -0.001), atol=1e-6)
    assert np.allclose(b.bvh["max"][0], (4.001, 4.001, 0.001), atol=1e-6)
    hits, st = O.trace_primary(b, scenes.reference_scene_camera(64, 64))
    assert int((hits["t"] < L.MAX_FLOAT).sum()) == 784 == int(st["hits"])
    assert abs(float(hits["t"][32, 32]) - 15.70128) < 1e-4
    n = 64 * 64
    assert round(st["pops"] / n, 1) == 9.7 and round(st["box_hits"] / n, 1) == 4.7
    assert round(st["leaf_tests"] / n, 1) == 0.8 and round(st["tri_tests"] / n, 1) == 0.4
    assert (hits["t"][hits["t"] >= L.MAX_FLOAT] == L.MAX_FLOAT).all()     # miss sentinel = (float)0x7F7FFFFF


# ---- a-2..a-5: the literal emulation of the reference kernels == stable sort -------------------------

def _sort_inputs(count, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "random":
        keys = rng.integers(0, 1 << 32, size=count, dtype=np.uint64).astype(np.uint32)
    elif kind == "morton_like":          # < 2^30 with many duplicates + 0xFFFFFFFF pads
        keys = (rng.integers(0, 1 << 30, size=count, dtype=np.uint64) >> 12 << 12).astype(np.uint32)
        keys[count - count // 5:] = F
    elif kind == "all_equal":
        keys = np.full(count, 0x12345678, dtype=np.uint32)
    else:
        keys = np.arange(count, dtype=np.uint32)[::-1].copy()
    vals = np.arange(count, dtype=np.uint32)
    vals[keys == F] = F if kind == "morton_like" else vals[keys == F]
    return keys, vals


@pytest.mark.parametrize("kind", ["random", "morton_like", "all_equal", "reversed"])
def test_literal_reference_sort_equals_stable_sort(kind):
    keys, vals = _sort_inputs(128 * 1024, 11, kind)          # 128 tiles of 1024
    lk, lv = O.sort_pairs_literal(keys, vals)
    sk, sv = O.sort_pairs(keys, vals)
    order = np.argsort(keys, kind="stable")
    assert (sk == keys[order]).all() and (sv == vals[order]).all()
    assert (lk == sk).all() and (lv == sv).all()


def test_literal_reference_sort_at_reference_capacity():
    """BLOCK_SIZE 512 x THREADS_PER_BLOCK 1024 = the reference's hard-wired 524 288 slots
    (Assets/_Scripts/Constants.cs:3-6)."""
    keys, vals = _sort_inputs(512 * 1024, 3, "morton_like")
    lk, lv = O.sort_pairs_literal(keys, vals)
    sk, sv = O.sort_pairs(keys, vals)
    assert (lk == sk).all() and (lv == sv).all()
    # the reference's own final check: monotone non-decreasing (ComputeBufferSorter.cs:150-177)
    assert (np.diff(sk.astype(np.int64)) >= 0).all()


def test_sort_edge_sizes():
    for count in (0, 1, 2, 255, 1023, 1025, 4097):
        keys, vals = _sort_inputs(count, count, "random")
        sk, sv = O.sort_pairs(keys, vals)
        order = np.argsort(keys, kind="stable")
        assert (sk == keys[order]).all() and (sv == vals[order]).all()


# ---- a-8 invariants ----------------------------------------------------------------------------------

def test_refit_boxes_enclose_children_and_nodes_are_covered():
    tris = scenes.random_triangles(3000, seed=9)
    b = O.Built(tris, capacity=scenes.capacity_for(3000))
    n = b.n
    # node coverage check of MeshBufferContainer.GetAllGpuData (:181-195)
    assert not ((b.leaf["index"][:n] == F) & (b.leaf["parent"][:n] == F)).any()
    assert not ((b.internal["index"][:n - 1] == F) & (b.internal["parent"][:n - 1] == F)).any()
    assert (b.internal["index"][:n - 1] == np.arange(n - 1)).all()
    assert (b.leaf["index"][:n] == np.arange(n)).all()
    assert b.internal["parent"][0] == F                           # root
    # every internal box is exactly the union of its children's boxes
    for i in range(n - 1):
        nd = b.internal[i]
        lb = b.bvh[nd["leftNode"]] if nd["leftNodeType"] == L.INTERNAL else b.triangle_aabb[b.indices[nd["leftNode"]]]
        rb = b.bvh[nd["rightNode"]] if nd["rightNodeType"] == L.INTERNAL else b.triangle_aabb[b.indices[nd["rightNode"]]]
        assert (b.bvh["min"][i] == np.minimum(lb["min"], rb["min"])).all()
        assert (b.bvh["max"][i] == np.maximum(lb["max"], rb["max"])).all()
    # untouched capacity slots keep the NullLeaf words (Assets/_Scripts/SceneDataTypes.cs:63-71)
    assert (b.internal.view(np.uint32).reshape(-1, 6)[n - 1:] == F).all()
    assert (b.leaf.view(np.uint32).reshape(-1, 2)[n:] == F).all()
    assert (b.keys[n:] == F).all() and (b.indices[n:] == F).all()


# ---- a-9 pieces ----------------------------------------------------------------------------------------

def test_slab_test_debug_fixture():
    """_debugRayBoxIntersectionTester: box (-5,-1,48)..(5,1,52), ray from (0,0,40)
    (Assets/__Scenes/Scene.unity:396-397, Assets/_Scripts/_debug/_debugRayBoxIntersectionTester.cs:17,33-45)."""
    bmin, bmax, origin = (-5.0, -1.0, 48.0), (5.0, 1.0, 52.0), (0.0, 0.0, 40.0)

    def hit(d):
        d = np.asarray(d, dtype=np.float32)
        with np.errstate(divide="ignore"):
            inv = (np.float32(1.0) / d).astype(np.float32)
        return O.ray_box(bmin, bmax, origin, inv)

    assert hit((0.0, 0.0, 1.0))                   # forward: red line
    assert not hit((0.0, 0.0, -1.0))              # box behind the ray: tmax < 0
    assert not hit((1.0, 0.0, 0.0)) and not hit((0.0, 1.0, 0.0))
    d = np.array([0.3, 0.0, 1.0]) / np.linalg.norm([0.3, 0.0, 1.0])
    assert hit(d)                                 # x reaches 2.4..3.6 inside the slab
    d = np.array([0.0, 0.2, 1.0]) / np.linalg.norm([0.0, 0.2, 1.0])
    assert not hit(d)                             # y = 1.6 at z = 48: above the box


def test_ray_generation_reference_camera():
    cam = scenes.reference_scene_camera(64, 64)
    o, d, inv = O.make_ray(cam, 32, 32)
    assert o.tolist() == [0.0, 0.0, np.float32(15.7)]
    assert d[2] < -0.999 and abs(np.linalg.norm(d) - 1) < 1e-6
    # +x on screen is -x in world for this camera (180 degree yaw), +y stays +y
    assert d[0] < 0 and d[1] > 0
    o2, d2, _ = O.make_ray(cam, 0, 0)
    assert d2[0] > 0 and d2[1] < 0


# ---- the C oracle against the fixtures of the independent literal emulation ---------------------------------

def _fixture(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if "triangles" in g.files:
        tris = np.ascontiguousarray(g["triangles"], dtype=L.TRIANGLE)
    elif name == "cfg1_4096":
        tris = scenes.random_triangles(4096, seed=1)
        assert (np.stack([tris["a"], tris["b"], tris["c"]], axis=1) == g["positions"]).all()
    else:
        tris = scenes.grid_scene()
    w, h = (int(x) for x in g["resolution"])
    cam = {"screen_width": w, "screen_height": h, "camera_fov": float(g["camera_fov"]), "near_plane": float(g["camera_near"]),
           "camera_to_world": g["camera_to_world"]}
    return tris, cam, g


@pytest.mark.parametrize("name", ["cfg1_4096", "grid_80x80", "example_object3", "viking_room"])
def test_oracle_matches_the_emulation_fixtures(name):
    """tests/golden/*.npz come from oracle/literal_emulation.py (thread-per-id Python emulation of the reference kernels,
    written independently of lbvh_oracle.c; generator: tests/golden/make_golden.py).  The C oracle reproduces every
    array bit for bit: Morton codes, sorted keys / indices, all node words, every box, every hit record field, the
    visit counters, and the shaded image of the textured asset."""
    tris, cam, g = _fixture(name)
    n = len(tris)
    b = O.Built(tris, capacity=scenes.capacity_for(n))
    assert (O.morton_aabb(tris)[0] == g["morton"]).all()
    assert (b.keys == g["sorted_keys"]).all() and (b.indices == g["sorted_indices"]).all()
    assert (b.internal[: n - 1].view(np.uint32).reshape(-1, 6) == g["internal"]).all()
    assert (b.leaf[:n].view(np.uint32).reshape(-1, 2) == g["leaf"]).all()
    assert (b.triangle_aabb["min"][:n].view(np.uint32) == g["tri_min"].view(np.uint32)).all()
    assert (b.triangle_aabb["max"][:n].view(np.uint32) == g["tri_max"].view(np.uint32)).all()
    assert (b.bvh["min"][: n - 1].view(np.uint32) == g["bvh_min"].view(np.uint32)).all()
    assert (b.bvh["max"][: n - 1].view(np.uint32) == g["bvh_max"].view(np.uint32)).all()
    hits, st = O.trace_primary(b, cam)
    assert (hits["t"].view(np.uint32) == g["hit_t"].view(np.uint32)).all() and (hits["tri"] == g["hit_tri"]).all()
    assert (hits["u"].view(np.uint32) == g["hit_u"].view(np.uint32)).all() and (hits["v"].view(np.uint32) == g["hit_v"].view(np.uint32)).all()
    assert [int(st[f]) for f in st.dtype.names] == g["stats"].tolist()
    if "shaded" in g.files:
        assert (O.shade(hits, b.triangles, g["texture"]).view(np.uint16) == g["shaded"]).all()


def test_reference_asset_is_the_procedural_grid():
    """Assets/_Assets/ExampleObject3.obj (the mesh wired into the reference scene, Scene.unity:364), parsed in the build
    container, is the 80x80 quad grid scenes.grid_scene() re-creates: same triangle count, same 1 156 distinct Morton
    codes, same 784 hits and visit counters from the reference camera."""
    a = np.load(os.path.join(GOLDEN, "example_object3.npz"))
    b = np.load(os.path.join(GOLDEN, "grid_80x80.npz"))
    assert len(a["triangles"]) == 12800 and len(np.unique(a["morton"])) == 1156
    assert a["stats"].tolist() == b["stats"].tolist() and int(a["stats"][4]) == 784
    assert (a["hit_t"] == b["hit_t"]).all()


def test_literal_emulation_equals_oracle_on_a_fresh_scene():
    """Not only on the frozen fixtures: a scene neither has seen, including duplicate Morton codes (a tiny mesh inside
    the +-125 box) and the refit threads run in random orders (any schedule gives the same boxes)."""
    from oracle import literal_emulation as E
    rng = np.random.default_rng(123)
    tris = scenes.random_triangles(300, seed=77, extent=0.4, edge=0.3)          # nearly all codes coincide
    r = E.awake(tris["a"], tris["b"], tris["c"], capacity=1024)
    b = O.Built(tris, capacity=1024)
    n = 300
    assert len(np.unique(r["morton"])) < 100                                                # 300 triangles, many ties
    assert (b.keys == r["sorted_keys"]).all() and (b.indices == r["sorted_indices"]).all()
    assert (b.internal.view(np.uint32).reshape(-1, 6) == r["internal"]).all()
    assert (b.leaf.view(np.uint32).reshape(-1, 2) == r["leaf"]).all()
    assert (b.bvh["min"][: n - 1] == r["bvh_min"]).all() and (b.bvh["max"][: n - 1] == r["bvh_max"]).all()
    for _ in range(3):
        mn, mx = E.bvh_constructor(n, r["internal"], r["leaf"], r["sorted_indices"], r["tri_min"], r["tri_max"],
                                   thread_order=rng.permutation(n))
        assert (mn == r["bvh_min"]).all() and (mx == r["bvh_max"]).all()
    cam = scenes.camera(20, 12, (0.1, 0.0, 1.5))
    t, tri, u, v, st = E.raytracing(r["scene"], cam)
    hits, ost = O.trace_primary(b, cam)
    assert (hits["t"] == t).all() and (hits["tri"] == tri).all() and (hits["u"] == u).all() and (hits["v"] == v).all()
    assert [int(ost[f]) for f in ost.dtype.names] == st
    # HLSL corner cases the emulation spells out
    assert E.clz32(0) == 32 and E.clz32(1) == 31 and E.clz32(0x80000000) == 0            # firstbithigh(0) = -1
    assert float(E.MAX_FLOAT) == 2139095040.0                                              # (float)0x7F7FFFFF


def test_openmp_build_equals_scalar_at_size():
    """the CPU baseline's OpenMP forms (parallel LSD sort, two-level DistributeKeys scan, flag hand-off refit) only switch
    on above 65 536 elements: same arrays as the serial restatement, bit for bit"""
    tris = scenes.tiled_torus(nu=40, nv=25, grid=4)            # 128 000 triangles, many equal Morton codes at this density
    b1 = O.Built(tris, capacity=scenes.capacity_for(len(tris)), threads=1)
    for threads in (3, 8):
        bt = O.Built(tris, capacity=scenes.capacity_for(len(tris)), threads=threads)
        assert (b1.keys == bt.keys).all() and (b1.indices == bt.indices).all()
        assert (b1.internal == bt.internal).all() and (b1.leaf == bt.leaf).all()
        assert (b1.bvh.view(np.uint32) == bt.bvh.view(np.uint32)).all()


def test_openmp_paths_equal_scalar():
    tris = scenes.random_triangles(2048, seed=4)
    b1 = O.Built(tris, threads=1)
    b4 = O.Built(tris, threads=4)
    assert (b1.keys == b4.keys).all() and (b1.internal == b4.internal).all() and (b1.leaf == b4.leaf).all()
    cam = scenes.camera(48, 32, (0.0, 0.0, 300.0))
    h1, s1 = O.trace_primary(b1, cam, threads=1)
    h4, s4 = O.trace_primary(b4, cam, threads=4)
    assert (h1 == h4).all() and s1 == s4


def test_openmp_forms_when_the_runtime_grants_fewer_threads_than_asked():
    """orc_sort_pairs_mt / orc_distribute_keys_mt computed their chunks from the REQUESTED thread count and
    indexed them by omp_get_thread_num(): with OMP_THREAD_LIMIT below the request whole chunks were never processed.
    Runs in a child process (the limit is read when libgomp loads) at a size that engages the parallel forms."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import oracle as O
rng = np.random.default_rng(11)
n = 200_000
keys = rng.integers(0, 1 << 30, n, dtype=np.uint64).astype(np.uint32)
keys[::7] = keys[3]                                  # duplicates: stability matters
vals = np.arange(n, dtype=np.uint32)
k1, v1 = O.sort_pairs(keys, vals)
k8, v8 = keys.copy(), vals.copy()
O._lib.orc_sort_pairs_mt(O._ptr(k8), O._ptr(v8), n, 8)
assert (k1 == k8).all() and (v1 == v8).all(), "parallel sort differs"
d1 = O.distribute_keys(k1, n)
d8 = k1.copy()
O._lib.orc_distribute_keys_mt(O._ptr(d8), n, 8)
assert (d1 == d8).all(), "parallel DistributeKeys differs"
print("ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    env = dict(os.environ, OMP_THREAD_LIMIT="2", OMP_DYNAMIC="true")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


# ---- a-9 tail: shading (Raytracing.compute:178-184) -------------------------------------------------------

def _shade_inputs(n_hits=300, seed=0):
    rng = np.random.default_rng(seed)
    tris = scenes.random_triangles(64, seed=3)
    for f in ("a_uv", "b_uv", "c_uv"):
        tris[f] = rng.uniform(-0.2, 1.2, (64, 2))          # some uv outside [0,1]: clamp addressing
    tex = rng.integers(0, 256, (16, 32, 4), dtype=np.uint8)
    hits = np.zeros(n_hits, dtype=L.HIT)
    hits["tri"] = rng.integers(0, 64, n_hits)
    hits["u"] = rng.uniform(0, 1, n_hits)
    hits["v"] = rng.uniform(0, 1, n_hits) * (1 - hits["u"])
    hits["t"] = np.where(rng.uniform(size=n_hits) < 0.5, 10.0, L.MAX_FLOAT)
    return tris, tex, hits


def test_shade_matches_an_independent_float32_recomputation():
    tris, tex, hits = _shade_inputs()
    img = O.shade(hits, tris, tex)
    f = np.float32
    out = np.zeros((len(hits), 4), np.float16)
    for i in range(len(hits)):
        t = tris[hits["tri"][i]]
        u, v = f(hits["u"][i]), f(hits["v"][i])
        w = f(f(1) - u) - v
        tu = f(f(w * t["a_uv"][0] + u * t["b_uv"][0]) + v * t["c_uv"][0])
        tv = f(f(w * t["a_uv"][1] + u * t["b_uv"][1]) + v * t["c_uv"][1])
        n = [f(f(w * t["a_normal"][k] + u * t["b_normal"][k]) + v * t["c_normal"][k]) for k in range(3)]
        l = f(0.57735026)                                   # scalar lightDir (Raytracing.compute:181)
        lam = max(f(0.4), f(f(l * n[0] + l * n[1]) + l * n[2]))
        x, y = f(tu * f(32) - f(0.5)), f(tv * f(16) - f(0.5))
        xf, yf = np.floor(x), np.floor(y)
        fx, fy = f(x - xf), f(y - yf)
        x0, x1 = int(min(max(xf, 0), 31)), int(min(max(xf + 1, 0), 31))
        y0, y1 = int(min(max(yf, 0), 15)), int(min(max(yf + 1, 0), 15))

        def c(yy, xx):
            return (tex[yy, xx].astype(f) / f(255)).astype(f)
        gx, gy = f(f(1) - fx), f(f(1) - fy)
        col = ((c(y0, x0) * gx + c(y0, x1) * fx).astype(f) * gy + (c(y1, x0) * gx + c(y1, x1) * fx).astype(f) * fy).astype(f)
        out[i, :3] = (col[:3] * lam).astype(f).astype(np.float16)
        out[i, 3] = 1.0 if hits["t"][i] != L.MAX_FLOAT else 0.0
    assert (img.view(np.uint16) == out.view(np.uint16)).all()


def test_shade_known_answers():
    # one white texel texture, normal (1,1,1)/sqrt(3): lambert = 0.57735026 * sum(n) = 1.0 -> colour 1, alpha by hit flag
    tris = np.zeros(1, dtype=L.TRIANGLE)
    nrm = np.float32(1 / np.sqrt(3))
    for f in ("a_normal", "b_normal", "c_normal"):
        tris[f] = (nrm, nrm, nrm)
    tex = np.full((1, 1, 4), 255, dtype=np.uint8)
    hits = np.zeros(2, dtype=L.HIT)
    hits["t"] = (5.0, L.MAX_FLOAT)
    img = O.shade(hits, tris, tex).astype(np.float32)
    assert np.allclose(img[0], (1, 1, 1, 1), atol=2e-3) and np.allclose(img[1], (1, 1, 1, 0), atol=2e-3)
    # a normal facing away is floored at 0.4 (max(0.4, .))
    for f in ("a_normal", "b_normal", "c_normal"):
        tris[f] = (-nrm, -nrm, -nrm)
    assert np.allclose(O.shade(hits, tris, tex).astype(np.float32)[0, :3], 0.4, atol=1e-3)


# ---- SURVEY 8(f) rank 3 (extension): dynamic scene + secondary rays ----------------------------------------

def test_animate_is_a_rigid_rotation_about_each_body_centre():
    tris, body, centres = scenes.tiled_torus(nu=8, nv=6, grid=2, with_bodies=True)
    same = O.animate(tris, body, centres, 0.0)
    # (p - c) + c is not exact in fp32, hence the tolerance; normals and uvs are
    assert np.allclose(same["a"], tris["a"], atol=1e-5) and (same["a_normal"] == tris["a_normal"]).all() and (same["a_uv"] == tris["a_uv"]).all()
    rot = O.animate(tris, body, centres, 0.3)
    ctr = centres[body][:, :3]
    for f in ("a", "b", "c"):
        assert np.allclose(np.linalg.norm(rot[f] - ctr, axis=1), np.linalg.norm(tris[f] - ctr, axis=1), atol=1e-4)
        assert np.allclose(rot[f][:, 1], tris[f][:, 1])                                  # Y axis
    assert np.allclose(np.linalg.norm(rot["b"] - rot["a"], axis=1), np.linalg.norm(tris["b"] - tris["a"], axis=1), atol=1e-4)
    assert not np.allclose(rot["a"], tris["a"])


def test_path_trace_is_deterministic_and_energy_bounded():
    tris = scenes.random_triangles(3000, seed=2, extent=40.0, edge=8.0)
    b = O.Built(tris, capacity=3072)
    cam = scenes.camera(40, 30, (0.0, 0.0, 120.0))
    img1, st1 = O.path_trace(b, cam, bounces=4, seed=7)
    img2, _ = O.path_trace(b, cam, bounces=4, seed=7, threads=4)
    img3, _ = O.path_trace(b, cam, bounces=4, seed=8)
    assert (img1.view(np.uint16) == img2.view(np.uint16)).all()                         # counter-based RNG
    assert not (img1.view(np.uint16) == img3.view(np.uint16)).all()
    hits, _ = O.trace_primary(b, cam)
    assert ((img1[..., 3] == 1) == (hits["t"] < L.MAX_FLOAT)).all()
    rgb = img1[..., :3].astype(np.float32)
    assert rgb.min() >= 0 and rgb.max() <= 1.0 + 1e-3                                   # albedo < 1, sky <= 1
    miss = hits["t"] >= L.MAX_FLOAT
    assert (rgb[miss].min(axis=1) >= 0.5 - 1e-3).all()                                  # primary misses see the sky
    # secondary rays never re-hit their own surface (t > t_min) and directions stay unit length
    st = O.path_begin(cam)
    O.path_scatter(b, hits.reshape(-1), st, 0, 7, 0.7)
    alive = st["alive"] == 1
    assert np.allclose(np.linalg.norm(st["dir"][alive], axis=1), 1.0, atol=1e-5)
    h2 = O.trace_rays(b, st, 1e-3)
    assert (h2["t"][alive & (h2["t"] < L.MAX_FLOAT)] > 1e-3).all()
    assert (h2["t"][~alive] == L.MAX_FLOAT).all()


def test_compose_known_answers():
    """ImageComposer.shader:49: lerp(col.rgb, colObject.rgb, colObject.a), alpha 1 — halves in, halves out."""
    bg = np.array([[0.25, 0.5, 1.0, 0.3], [0.125, 0.75, 0.0, 1.0], [1.0, 1.0, 1.0, 1.0]], np.float16)
    ob = np.array([[1.0, 0.0, 0.5, 0.5], [0.5, 0.5, 0.5, 0.0], [0.0, 0.25, 0.5, 1.0]], np.float16)
    out = O.compose(bg, ob)
    assert out.tolist() == [[0.625, 0.25, 0.75, 1.0], [0.125, 0.75, 0.0, 1.0], [0.0, 0.25, 0.5, 1.0]]
    # subnormal and large halves survive the half -> float -> half trip
    x = np.array([[6e-8, 65504.0, -2.0, 1.0]], np.float16)
    assert (O.compose(x, np.zeros((1, 4), np.float16))[0, :3] == x[0, :3]).all()

