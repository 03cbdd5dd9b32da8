"""ctypes wrapper of the CPU oracle (oracle/liblbvh_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the
product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from unitysimpleraytracing_amd import layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblbvh_oracle.so")


def build():
    """gcc only (`make` = liblbvh_oracle.so): the CPU checker does not need the GPU toolchain; the rocPRIM cross-check of the
    sort is `make gpu-checkers`, built by __graft_entry__.build()."""
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


if not os.path.exists(LIB_PATH):
    build()

# LBVH_ORACLE_LIB: another build of the same oracle (the sanitizer job loads liblbvh_oracle_asan.so / _tsan.so)
_lib = C.CDLL(os.environ.get("LBVH_ORACLE_LIB") or LIB_PATH)
_P, _U32, _I32 = C.c_void_p, C.c_uint32, C.c_int32
_F3 = C.POINTER(C.c_float)


class _Camera(C.Structure):
    _fields_ = [("screen_width", _I32), ("screen_height", _I32), ("camera_fov", C.c_float),
                ("near_plane", C.c_float), ("camera_to_world", C.c_float * 16)]


class _Scene(C.Structure):
    _fields_ = [("n", _U32), ("sorted_indices", _P), ("triangle_aabb", _P), ("internal_nodes", _P),
                ("leaf_nodes", _P), ("bvh", _P), ("triangles", _P)]


_lib.orc_num_threads.restype = _I32
_lib.orc_morton_aabb.argtypes = [_P, _U32, _U32, _F3, _F3, _P, _P, _P, _I32]
_lib.orc_morton_aabb.restype = None
_lib.orc_sort_pairs.argtypes = [_P, _P, _U32]
_lib.orc_sort_pairs.restype = None
_lib.orc_sort_pairs_literal.argtypes = [_P, _P, _U32]
_lib.orc_sort_pairs_literal.restype = _I32
_lib.orc_distribute_keys.argtypes = [_P, _U32]
_lib.orc_distribute_keys.restype = None
_lib.orc_build_tree.argtypes = [_U32, _P, _P, _P, _I32]
_lib.orc_build_tree.restype = _I32
_lib.orc_refit.argtypes = [_U32, _P, _P, _P, _P, _P]
_lib.orc_refit.restype = _I32
_lib.orc_ray_box.argtypes = [_F3, _F3, _F3, _F3]
_lib.orc_ray_box.restype = _I32
_lib.orc_make_ray.argtypes = [C.POINTER(_Camera), _U32, _U32, _F3, _F3, _F3]
_lib.orc_make_ray.restype = None
_lib.orc_trace_primary.argtypes = [C.POINTER(_Camera), _I32, _I32, _I32, _I32, _I32, _I32,
                                   C.POINTER(_Scene), _P, _P, _I32]
_lib.orc_trace_primary.restype = _I32
_lib.orc_trace_primary_rule.argtypes = [C.POINTER(_Camera), _I32, _I32, _I32, _I32, _I32, _I32,
                                        C.POINTER(_Scene), _P, _P, _I32, _I32]
_lib.orc_trace_primary_rule.restype = _I32
_lib.orc_build_all.argtypes = [_P, _U32, _U32, _F3, _F3, _P, _P, _P, _P, _P, _P, _I32]
_lib.orc_build_all.restype = _I32


_lib.orc_shade.argtypes = [_P, C.c_size_t, _P, _P, _I32, _I32, _P]
_lib.orc_shade.restype = None


def shade(hits, triangles, texture_rgba8):
    """hits: array of layouts.HIT (any shape); texture: (h, w, 4) uint8, row 0 at v = 0.
    Returns float16 array hits.shape + (4,)."""
    h = np.ascontiguousarray(hits, dtype=L.HIT)
    tris = np.ascontiguousarray(triangles, dtype=L.TRIANGLE)
    tex = np.ascontiguousarray(texture_rgba8, dtype=np.uint8)
    out = np.zeros(h.shape + (4,), dtype=np.uint16)
    _lib.orc_shade(_ptr(h), h.size, _ptr(tris), _ptr(tex), tex.shape[1], tex.shape[0], _ptr(out))
    return out.view(np.float16)


_lib.orc_compose.argtypes = [_P, _P, C.c_size_t, _P]
_lib.orc_compose.restype = None


def compose(background, obj):
    """ImageComposer: float16 arrays (..., 4) -> float16 (..., 4)."""
    b = np.ascontiguousarray(background, dtype=np.float16)
    o = np.ascontiguousarray(obj, dtype=np.float16)
    out = np.zeros(b.shape, dtype=np.uint16)
    _lib.orc_compose(_ptr(b), _ptr(o), b.size // 4, _ptr(out))
    return out.view(np.float16)


_lib.orc_animate.argtypes = [_P, _U32, _P, _P, C.c_float, C.c_float, _P]
_lib.orc_animate.restype = None
_lib.orc_path_begin.argtypes = [C.POINTER(_Camera), _P]
_lib.orc_path_begin.restype = None
_lib.orc_trace_rays.argtypes = [_P, C.c_size_t, C.c_float, C.POINTER(_Scene), _P, _I32]
_lib.orc_trace_rays.restype = _I32
_lib.orc_path_scatter.argtypes = [C.POINTER(_Scene), _P, C.c_size_t, _U32, _U32, C.c_float, _P]
_lib.orc_path_scatter.restype = None
_lib.orc_path_resolve.argtypes = [_P, C.c_size_t, _P]
_lib.orc_path_resolve.restype = None


def animate(rest, body, centres, angle):
    rest = np.ascontiguousarray(rest, dtype=L.TRIANGLE)
    out = np.zeros_like(rest)
    _lib.orc_animate(_ptr(rest), len(rest), _ptr(np.ascontiguousarray(body, dtype=np.uint32)),
                     _ptr(np.ascontiguousarray(centres, dtype=np.float32)),
                     float(np.float32(np.cos(angle))), float(np.float32(np.sin(angle))), _ptr(out))
    return out


def path_begin(camera):
    cam = _camera(camera)
    st = np.zeros(cam.screen_width * cam.screen_height, dtype=L.PATH_STATE)
    _lib.orc_path_begin(C.byref(cam), _ptr(st))
    return st


def trace_rays(built, states, t_min, threads=1):
    hits = np.zeros(len(states), dtype=L.HIT)
    s = built.scene()
    rc = _lib.orc_trace_rays(_ptr(states), len(states), float(t_min), C.byref(s), _ptr(hits), threads)
    if rc != 0:
        raise ValueError(f"orc_trace_rays rc={rc}")
    return hits


def path_scatter(built, hits, states, bounce, seed, albedo):
    s = built.scene()
    _lib.orc_path_scatter(C.byref(s), _ptr(np.ascontiguousarray(hits, dtype=L.HIT)), len(states), int(bounce), int(seed),
                          float(albedo), _ptr(states))
    return states


def path_resolve(states):
    out = np.zeros((len(states), 4), dtype=np.uint16)
    _lib.orc_path_resolve(_ptr(states), len(states), _ptr(out))
    return out.view(np.float16)


def path_trace(built, camera, bounces=4, t_min=1e-3, albedo=0.7, seed=1, threads=1):
    """The DynamicPathTracer.render pipeline on the host.  Every segment — the primary rays too — takes the closest hit with
    exact ties going to the lowest triangle index (orc_trace_rays: the order-independent rule of LBVH_TRACE_FAST and of the GPU's
    per-ray walkers), so the whole path state is comparable bit for bit; the primary rays accept any t (no t_min: Raytracing.compute
    has no t > 0 test)."""
    st = path_begin(camera)
    hits = trace_rays(built, st, -3.0e38, threads=threads)
    path_scatter(built, hits.reshape(-1), st, 0, seed, albedo)
    for b in range(1, bounces + 1):
        h = trace_rays(built, st, t_min, threads=threads)
        path_scatter(built, h, st, b, seed, albedo)
    cam = _camera(camera)
    return path_resolve(st).reshape(cam.screen_height, cam.screen_width, 4), st


def _ptr(a):
    return a.ctypes.data_as(_P)


def _f3(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(_F3)


def num_threads():
    return int(_lib.orc_num_threads())


def _camera(d):
    cam = _Camera()
    cam.screen_width, cam.screen_height = d["screen_width"], d["screen_height"]
    cam.camera_fov, cam.near_plane = d["camera_fov"], d["near_plane"]
    for i, v in enumerate(np.asarray(d["camera_to_world"], dtype=np.float32).reshape(-1)):
        cam.camera_to_world[i] = float(v)
    return cam


def morton_aabb(triangles, capacity=None, box_min=L.SCENE_BOX_MIN, box_max=L.SCENE_BOX_MAX, threads=1):
    tris = np.ascontiguousarray(triangles, dtype=L.TRIANGLE)
    n = len(tris)
    cap = n if capacity is None else int(capacity)
    keys = np.empty(cap, dtype=np.uint32)
    idx = np.empty(cap, dtype=np.uint32)
    aabb = np.zeros(cap, dtype=L.AABB)
    bmin = np.ascontiguousarray(box_min, dtype=np.float32)
    bmax = np.ascontiguousarray(box_max, dtype=np.float32)
    _lib.orc_morton_aabb(_ptr(tris), n, cap, bmin.ctypes.data_as(_F3), bmax.ctypes.data_as(_F3),
                         _ptr(keys), _ptr(idx), _ptr(aabb), threads)
    return keys, idx, aabb


def sort_pairs(keys, values):
    k = np.array(keys, dtype=np.uint32, copy=True)
    v = np.array(values, dtype=np.uint32, copy=True)
    _lib.orc_sort_pairs(_ptr(k), _ptr(v), len(k))
    return k, v


def sort_pairs_literal(keys, values):
    k = np.array(keys, dtype=np.uint32, copy=True)
    v = np.array(values, dtype=np.uint32, copy=True)
    assert len(k) % 1024 == 0
    rc = _lib.orc_sort_pairs_literal(_ptr(k), _ptr(v), len(k) // 1024)
    if rc != 0:
        raise ValueError(f"orc_sort_pairs_literal rc={rc}")
    return k, v


def distribute_keys(keys, n):
    k = np.array(keys, dtype=np.uint32, copy=True)
    _lib.orc_distribute_keys(_ptr(k), int(n))
    return k


def build_tree(sorted_keys, n, capacity=None, threads=1):
    cap = int(n) if capacity is None else int(capacity)
    internal = np.empty(cap, dtype=L.INTERNAL_NODE)
    leaf = np.empty(cap, dtype=L.LEAF_NODE)
    internal.view(np.uint32)[:] = L.NULL
    leaf.view(np.uint32)[:] = L.NULL
    k = np.ascontiguousarray(sorted_keys, dtype=np.uint32)
    rc = _lib.orc_build_tree(int(n), _ptr(k), _ptr(internal), _ptr(leaf), threads)
    if rc != 0:
        raise ValueError(f"orc_build_tree rc={rc}")
    return internal, leaf


def refit(n, internal, leaf, triangle_aabb, sorted_indices, capacity=None):
    cap = int(n) if capacity is None else int(capacity)
    bvh = np.zeros(cap, dtype=L.AABB)
    rc = _lib.orc_refit(int(n), _ptr(np.ascontiguousarray(internal)), _ptr(np.ascontiguousarray(leaf)),
                        _ptr(np.ascontiguousarray(triangle_aabb)),
                        _ptr(np.ascontiguousarray(sorted_indices, dtype=np.uint32)), _ptr(bvh))
    if rc != 0:
        raise ValueError(f"orc_refit rc={rc}")
    return bvh


def ray_box(bmin, bmax, origin, inv_dir):
    return bool(_lib.orc_ray_box(_f3(bmin), _f3(bmax), _f3(origin), _f3(inv_dir)))


def ray_triangle(origin, direction, triangle):
    """RayTriangleIntersection (Raytracing.compute:37-73) of one ray with one TRIANGLE record: its t, or MAX_FLOAT"""
    _lib.orc_ray_triangle.restype = C.c_float
    return np.float32(_lib.orc_ray_triangle(_f3(origin), _f3(direction), _f3(triangle["a"]), _f3(triangle["b"]), _f3(triangle["c"])))


def make_ray(camera, px, py):
    cam = _camera(camera)
    o = np.zeros(3, dtype=np.float32)
    d = np.zeros(3, dtype=np.float32)
    i = np.zeros(3, dtype=np.float32)
    _lib.orc_make_ray(C.byref(cam), px, py, o.ctypes.data_as(_F3), d.ctypes.data_as(_F3), i.ctypes.data_as(_F3))
    return o, d, i


def box_entry(bmin, bmax, origin, inv_dir):
    """tmin of RayBoxIntersection (Raytracing.compute:75-87) in strict fp32, operation by operation: the distance at which the
    slab test says the ray enters the box."""
    f = np.float32
    o, i = np.asarray(origin, dtype=f), np.asarray(inv_dir, dtype=f)
    with np.errstate(all="ignore"):
        t1 = (np.asarray(bmin, dtype=f) - o) * i
        t2 = (np.asarray(bmax, dtype=f) - o) * i
        return f(np.fmax(np.fmax(np.fmin(t1[0], t2[0]), np.fmin(t1[1], t2[1])), np.fmin(t1[2], t2[2])))


def winner_before_its_box(built, camera, px, py, hit):
    """Is this hit record an fp32 ARTEFACT of the reference's triangle test: a t that lies BEFORE the distance at which the ray
    enters the winning triangle's own padded AABB (Raytracing.compute:37-73 on a ray within a fraction of a degree of the
    triangle's plane: det ~ 1e-4, the dot products cancel, t is noise; in double precision such a ray usually misses).  The
    reference, which prunes nothing, reports such a t whenever the ray's line passes the triangle's box; a walker that skips boxes
    entered beyond the best hit so far reports it only if it happens to visit that leaf before a genuine hit behind it.  This is
    the ONLY way LBVH_TRACE_FAST / _FAST_EXACT can differ from the reference: t_fast > t_reference and this
    predicate true for the reference's winner."""
    if not float(hit["t"]) < float(L.MAX_FLOAT):
        return False
    o, d, inv = make_ray(camera, int(px), int(py))
    box = built.triangle_aabb[int(hit["tri"])]
    return bool(np.float32(hit["t"]) < box_entry(box["min"], box["max"], o, inv))


def unexplained_mismatches(built, camera, reference_hits, fast_hits, origin=(0, 0), words=False):
    """Pixels (y, x) where a fast-mode frame differs from the reference's (in t, or in any word with words=True) and the difference
    is NOT the fp32 artefact above.  `origin` = (x0, y0) of the frames' top-left pixel.  Returns (unexplained, explained)."""
    if words:
        a = np.ascontiguousarray(reference_hits).view(np.uint32).reshape(reference_hits.shape + (4,))
        b = np.ascontiguousarray(fast_hits).view(np.uint32).reshape(fast_hits.shape + (4,))
        bad = np.argwhere((a != b).any(axis=-1))
    else:
        bad = np.argwhere(fast_hits["t"] != reference_hits["t"])
    unexplained, explained = [], []
    for y, x in bad:
        r, f = reference_hits[y, x], fast_hits[y, x]
        ok = float(f["t"]) > float(r["t"]) and winner_before_its_box(built, camera, x + origin[0], y + origin[1], r)
        (explained if ok else unexplained).append((int(y), int(x)))
    return unexplained, explained


class Built:
    """All seven scene arrays of one Awake() build on the host."""

    def __init__(self, triangles, capacity=None, threads=1):
        tris = np.ascontiguousarray(triangles, dtype=L.TRIANGLE)
        n = len(tris)
        cap = n if capacity is None else int(capacity)
        self.n, self.capacity = n, cap
        self.triangles = np.zeros(cap, dtype=L.TRIANGLE)
        self.triangles[:n] = tris
        self.keys = np.empty(cap, dtype=np.uint32)
        self.indices = np.empty(cap, dtype=np.uint32)
        self.triangle_aabb = np.zeros(cap, dtype=L.AABB)
        self.internal = np.empty(cap, dtype=L.INTERNAL_NODE)
        self.leaf = np.empty(cap, dtype=L.LEAF_NODE)
        self.bvh = np.zeros(cap, dtype=L.AABB)
        self.rebuild(threads)

    def rebuild(self, threads=1):
        """the whole Awake() chain again into the same arrays (CPU baseline: warm pages); returns the seconds it took"""
        import time
        t0 = time.perf_counter()
        rc = _lib.orc_build_all(_ptr(self.triangles), self.n, self.capacity, _f3(L.SCENE_BOX_MIN), _f3(L.SCENE_BOX_MAX),
                                _ptr(self.keys), _ptr(self.indices), _ptr(self.triangle_aabb),
                                _ptr(self.internal), _ptr(self.leaf), _ptr(self.bvh), threads)
        self.build_seconds = time.perf_counter() - t0
        if rc != 0:
            raise ValueError(f"orc_build_all rc={rc}")
        return self.build_seconds

    def scene(self):
        s = _Scene()
        s.n = self.n
        s.sorted_indices = self.indices.ctypes.data
        s.triangle_aabb = self.triangle_aabb.ctypes.data
        s.internal_nodes = self.internal.ctypes.data
        s.leaf_nodes = self.leaf.ctypes.data
        s.bvh = self.bvh.ctypes.data
        s.triangles = self.triangles.ctypes.data
        return s


def trace_primary(built, camera, rect=None, step=(1, 1), threads=1, fast_rule=False):
    """Returns (hits[h, w] of layouts.HIT, stats of layouts.TRACE_STATS) for the sampled grid.  fast_rule: the reference's loop
    with the accept rule of the library's fast modes (a computed t in front of its own triangle's box does not count):
    what LBVH_TRACE_FAST_EXACT returns word for word and LBVH_TRACE_FAST in t — the reference's frame except where the
    reference's winner is such a t (winner_before_its_box)."""
    cam = _camera(camera)
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, cam.screen_width, cam.screen_height)
    sx, sy = step
    w = (x1 - x0 + sx - 1) // sx
    h = (y1 - y0 + sy - 1) // sy
    hits = np.zeros((h, w), dtype=L.HIT)
    stats = np.zeros(1, dtype=L.TRACE_STATS)
    s = built.scene()
    rc = _lib.orc_trace_primary_rule(C.byref(cam), x0, y0, x1, y1, sx, sy, C.byref(s), _ptr(hits), _ptr(stats), threads, 1 if fast_rule else 0)
    if rc != 0:
        raise ValueError(f"orc_trace_primary rc={rc}")
    return hits, stats[0]
