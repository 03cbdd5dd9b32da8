"""literal_emulation.py — a SECOND, independent restatement of the reference hot path: plain Python, one
emulated GPU thread per dispatch-thread id, written from the HLSL / C# text alone (not from lbvh_oracle.c).
TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/: only tests/, smoke() and bench.py's CPU leg may
import it; here only tests/golden/make_golden.py and the CPU tests do).

Why it exists: the reference holds no golden vectors and cannot run in this image (HLSL SM6 via DXC + Unity C#), so
parity is "unpinned" by the letter.  The strongest evidence available is two restatements written independently —
this file (thread-per-id emulation, Python integers with explicit 32-bit wrap-around, numpy float32 scalars that
round after every operation) and oracle/lbvh_oracle.c (C, semantic loops) — agreeing bit for bit on every array.
tests/golden/*.npz are generated from THIS file; the C oracle and the GPU library are both checked against them.

What is emulated, with the reference lines (Sc/ = Assets/_Scripts/, Sh/ = Assets/_Shaders/):
  morton_aabb          Sc/MeshBufferContainer.cs:32-83, 123-146  (the C# loop, float fields => every op rounds to fp32)
  distribute_keys      Sc/MeshBufferContainer.cs:154-169
  tree_constructor     Sh/BVH/BVH.compute:72-203   kernel TreeConstructor: clz32 via firstbithigh (firstbithigh(0) = -1,
                       in uint arithmetic 31 - 0xFFFFFFFF = 32), delta, DetermineRange with HLSL's int -> uint promotion
                       in `idx + lmax * d` and `idx + (l + t) * d` (:94, :99) wrapped back to int, FindSplit
  bvh_constructor      Sh/BVH/BVH.compute:206-274  kernel BVHConstructor: one thread per leaf, InterlockedCompareExchange
                       on atomicsData (first arrival leaves, second merges), threads run in any order given
  raytracing           Sh/Raytracing/Raytracing.compute:105-176 (+ :37-103): ray generation, the 64-entry stack loop
                       (push left, push right, pop), CheckTriangle, strict `<`, MAX_FLOAT = (float)0x7F7FFFFF
  shade                Sh/Raytracing/Raytracing.compute:178-184
The sort is taken as its contract (stable by key; numpy's stable argsort): the reference's five sort kernels are
emulated literally in C (orc_sort_pairs_literal) and pinned equal to the stable sort in tests/test_oracle_kat.py.

Conventions the HLSL text leaves to the compiler, fixed here exactly as DESIGN.md section 2 lists them (both
restatements and the GPU library use the same): dot products and matrix rows summed left to right with every product
and sum rounded to fp32 (no FMA contraction); normalize(v) = v / sqrt(dot(v, v)); 1 / x and a / b are IEEE divisions;
min / max return the non-NaN operand (fminf / fmaxf).  North_star's 1e-5 tolerance on hit t covers a DXC build that
chooses otherwise.
"""
import numpy as np

f32 = np.float32
U32 = 0xFFFFFFFF
INTERNAL_NODE, LEAF_NODE = 0, 1                      # Sh/Constants.cginc:17-18
MAX_FLOAT = f32(0x7F7FFFFF)                          # Sh/Constants.cginc:7: an INTEGER literal converted to float


def _u32(x):
    return int(x) & U32


def _i32(x):
    """reinterpret a 32-bit pattern as HLSL int"""
    x = int(x) & U32
    return x - (1 << 32) if x & 0x80000000 else x


# ---------------------------------------------------------------------------------------------------------------
# Sc/MeshBufferContainer.cs
# ---------------------------------------------------------------------------------------------------------------
def expand_bits(v):                                   # :32-39
    v = _u32(v * 0x00010001) & 0xFF0000FF
    v = _u32(v * 0x00000101) & 0x0F00F00F
    v = _u32(v * 0x00000011) & 0xC30C30C3
    v = _u32(v * 0x00000005) & 0x49249249
    return v


def morton3d(x, y, z):                                # :41-50
    q = []
    for c in (x, y, z):
        c = f32(c) * f32(1024.0)
        c = min(max(c, f32(0.0)), f32(1023.0))        # Math.Min(Math.Max(.., 0), 1023)
        q.append(int(c))                              # (uint) truncation
    return _u32(expand_bits(q[0]) * 4 + expand_bits(q[1]) * 2 + expand_bits(q[2]))


def morton_aabb(a, b, c, size=125.0):
    """The constructor loop :123-146 for triangles given as (n, 3) float32 vertex arrays.
    Returns keys[n] (u32), aabb_min[n, 3], aabb_max[n, 3] (f32)."""
    n = len(a)
    keys = np.zeros(n, dtype=np.uint32)
    mn = np.zeros((n, 3), dtype=np.float32)
    mx = np.zeros((n, 3), dtype=np.float32)
    whole_min, whole_max = f32(-1.0) * f32(size), f32(size)          # :11-15
    for i in range(n):
        cen = []
        for k in range(3):
            lo = min(min(f32(a[i][k]), f32(b[i][k])), f32(c[i][k])) - f32(0.001)     # :54-58
            hi = max(max(f32(a[i][k]), f32(b[i][k])), f32(c[i][k])) + f32(0.001)     # :59-63
            mn[i][k], mx[i][k] = lo, hi
            ce = (lo + hi) * f32(0.5)                                                # :65
            ce = ce - whole_min                                                      # :76-78
            ce = ce / (whole_max - whole_min)                                        # :79-81
            cen.append(ce)
        keys[i] = morton3d(cen[0], cen[1], cen[2])                                   # :130
    return keys, mn, mx


def distribute_keys(keys, n):                         # :154-169 (uint arithmetic)
    out = [int(k) for k in keys]
    new_current = 0
    old_current = out[0]
    out[0] = new_current
    for i in range(1, n):
        new_current = _u32(new_current + max(_u32(out[i] - old_current), 1))
        old_current = out[i]
        out[i] = new_current
    return np.array(out, dtype=np.uint32)


# ---------------------------------------------------------------------------------------------------------------
# Sh/BVH/BVH.compute — kernel TreeConstructor
# ---------------------------------------------------------------------------------------------------------------
def firstbithigh(v):
    """HLSL firstbithigh on uint: index of the highest set bit, -1 (0xFFFFFFFF) for 0"""
    v = _u32(v)
    return v.bit_length() - 1 if v else -1


def clz32(v):                                         # :72-75   `31 - firstbithigh(v)` in uint arithmetic
    return _u32(31 - _u32(firstbithigh(v)))


class _TreeThread:
    """one dispatch thread of TreeConstructor; `codes` = sortedMortonCodes"""

    def __init__(self, codes, num_objects):
        self.codes, self.num = codes, num_objects

    def delta(self, x, y):                            # :77-87   (int x, int y)
        if 0 <= x <= self.num - 1 and 0 <= y <= self.num - 1:
            return _i32(clz32(int(self.codes[x]) ^ int(self.codes[y])))
        return -1

    def determine_range(self, idx):                   # :89-106
        diff = self.delta(idx, idx + 1) - self.delta(idx, idx - 1)
        d = (diff > 0) - (diff < 0)                                                   # sign()
        dmin = self.delta(idx, idx - d)
        lmax = 2                                                                      # uint
        # `idx + lmax * d`: int * uint promotes to uint, the sum is uint, the int parameter reinterprets it
        while self.delta(idx, _i32(_u32(idx) + _u32(lmax * _u32(d)))) > dmin:
            lmax = _u32(lmax * 2)
        l = 0                                                                         # int
        t = lmax // 2                                                                 # uint
        while t >= 1:
            if self.delta(idx, _i32(_u32(idx) + _u32(_u32(_u32(l) + t) * _u32(d)))) > dmin:
                l = _i32(_u32(l) + t)
            t //= 2
        j = idx + l * d
        return min(idx, j), max(idx, j)

    def find_split(self, first, last):                # :108-146
        first_code, last_code = int(self.codes[first]), int(self.codes[last])
        if first_code == last_code:
            return (first + last) >> 1
        common_prefix = _i32(clz32(first_code ^ last_code))
        split = first
        step = last - first
        while True:
            step = (step + 1) >> 1
            new_split = split + step
            if new_split < last:
                split_prefix = _i32(clz32(first_code ^ int(self.codes[new_split])))
                if split_prefix > common_prefix:
                    split = new_split
            if not step > 1:
                break
        return split


def tree_constructor(codes, triangles_count, capacity=None):
    """Runs every thread id of the dispatch; returns internal[capacity, 6] and leaf[capacity, 2] u32 arrays that start
    as NullLeaf (every word 0xFFFFFFFF, Sc/SceneDataTypes.cs:63-71, 85-89).  Columns of internal: leftNode,
    leftNodeType, rightNode, rightNodeType, parent, index (Sh/Constants.cginc:20-28); of leaf: parent, index."""
    cap = triangles_count if capacity is None else capacity
    internal = np.full((cap, 6), U32, dtype=np.uint32)
    leaf = np.full((cap, 2), U32, dtype=np.uint32)
    th = _TreeThread(codes, int(triangles_count))
    for thread_id in range(cap):
        if not thread_id < _u32(triangles_count - 1):                                 # :155 (uint compare)
            continue
        first, last = th.determine_range(thread_id)
        split = th.find_split(first, last)
        internal[thread_id][5] = thread_id                                            # :165 .index
        if split == first:                                                            # :168-177
            leaf[split] = (thread_id, split)
            internal[thread_id][0] = split
            internal[thread_id][1] = LEAF_NODE
        else:                                                                         # :178-183
            internal[split][4] = thread_id
            internal[thread_id][0] = split
            internal[thread_id][1] = INTERNAL_NODE
        if split + 1 == last:                                                         # :186-195
            leaf[split + 1] = (thread_id, split + 1)
            internal[thread_id][2] = split + 1
            internal[thread_id][3] = LEAF_NODE
        else:                                                                         # :196-201
            internal[split + 1][4] = thread_id
            internal[thread_id][2] = split + 1
            internal[thread_id][3] = INTERNAL_NODE
    return internal, leaf


# ---------------------------------------------------------------------------------------------------------------
# Sh/BVH/BVH.compute — kernel BVHConstructor
# ---------------------------------------------------------------------------------------------------------------
def _merge(lmin, lmax, rmin, rmax):                   # MergeAABB :206-224
    return (np.array([min(lmin[k], rmin[k]) for k in range(3)], dtype=np.float32),
            np.array([max(lmax[k], rmax[k]) for k in range(3)], dtype=np.float32))


def bvh_constructor(triangles_count, internal, leaf, sorted_indices, tri_min, tri_max, thread_order=None):
    """One thread per leaf (:233) walking towards the root; atomicsData starts at 0 (Sc/BVHConstructor.cs:41).
    `thread_order`: the order in which the emulated threads run to completion (default: by id; any permutation is a
    schedule the GPU could produce).  Returns bvh_min, bvh_max [n - 1, 3]."""
    n = int(triangles_count)
    atomics = [0] * n
    bmin = np.zeros((n, 3), dtype=np.float32)
    bmax = np.zeros((n, 3), dtype=np.float32)
    order = range(n) if thread_order is None else thread_order
    for thread_id in order:
        parent = int(leaf[thread_id][0])                                              # :235
        while parent != U32:                                                          # :236
            old = atomics[parent]                                                     # InterlockedCompareExchange(.., 0, 1, old)
            if old == 0:
                atomics[parent] = 1
                break                                                                 # :240-243
            left_id, left_type, right_id, right_type = (int(x) for x in internal[parent][:4])
            if left_type == INTERNAL_NODE:                                            # :251-258
                lmn, lmx = bmin[left_id], bmax[left_id]
            else:
                t = int(sorted_indices[left_id])
                lmn, lmx = tri_min[t], tri_max[t]
            if right_type == INTERNAL_NODE:                                           # :260-267
                rmn, rmx = bmin[right_id], bmax[right_id]
            else:
                t = int(sorted_indices[right_id])
                rmn, rmx = tri_min[t], tri_max[t]
            bmin[parent], bmax[parent] = _merge(lmn, lmx, rmn, rmx)                   # :269
            parent = int(internal[parent][4])                                         # :271
    return bmin[: n - 1], bmax[: n - 1]


# ---------------------------------------------------------------------------------------------------------------
# Sh/Raytracing/Raytracing.compute
# ---------------------------------------------------------------------------------------------------------------
def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def ray_triangle_intersection(orig, direction, v0, v1, v2):        # :37-73 -> (distance, u, v)
    e1, e2 = _sub(v1, v0), _sub(v2, v0)
    pvec = _cross(direction, e2)
    det = _dot(e1, pvec)
    if det < f32(1e-8) and det > f32(-1e-8):
        return MAX_FLOAT, f32(0), f32(0)
    inv_det = f32(1) / det
    tvec = _sub(orig, v0)
    u = _dot(tvec, pvec) * inv_det
    if u < 0 or u > 1:
        return MAX_FLOAT, f32(0), f32(0)
    qvec = _cross(tvec, e1)
    v = _dot(direction, qvec) * inv_det
    if v < 0 or u + v > 1:
        return MAX_FLOAT, f32(0), f32(0)
    return _dot(e2, qvec) * inv_det, u, v


def _fmin(a, b):
    return b if a != a else (a if b != b else (a if a < b else b))


def _fmax(a, b):
    return b if a != a else (a if b != b else (a if a > b else b))


def _slabs(bmin, bmax, origin, inv_dir):                            # :77-84: tmin, tmax
    with np.errstate(invalid="ignore", over="ignore"):
        t1 = [(bmin[k] - origin[k]) * inv_dir[k] for k in range(3)]
        t2 = [(bmax[k] - origin[k]) * inv_dir[k] for k in range(3)]
    tmin1 = [_fmin(t1[k], t2[k]) for k in range(3)]
    tmax1 = [_fmax(t1[k], t2[k]) for k in range(3)]
    return _fmax(tmin1[0], _fmax(tmin1[1], tmin1[2])), _fmin(tmax1[0], _fmin(tmax1[1], tmax1[2]))


def ray_box_intersection(bmin, bmax, origin, inv_dir):              # :75-87
    tmin, tmax = _slabs(bmin, bmax, origin, inv_dir)
    return bool(tmax > tmin and tmax > 0)


def ray_box_entry(bmin, bmax, origin, inv_dir):
    """tmin of the slab test: the distance at which it says the ray enters the box (not a reference function: the accept rule of
    the library's fast modes compares a triangle's computed t with it)"""
    return _slabs(bmin, bmax, origin, inv_dir)[0]


def make_ray(cam, idx, idy):                                        # :108-126
    near, fov = f32(cam["near_plane"]), f32(cam["camera_fov"])
    sw, sh = f32(cam["screen_width"]), f32(cam["screen_height"])
    height = f32(2) * near * fov
    width = sw * height / sh
    d = (-width / f32(2) + width / sw * (f32(idx) + f32(0.5)),
         -height / f32(2) + height / sh * (f32(idy) + f32(0.5)),
         -near)
    m = [f32(x) for x in np.asarray(cam["camera_to_world"], dtype=np.float32).reshape(-1)]
    o4, d4 = (f32(0), f32(0), f32(0), f32(1)), (d[0], d[1], d[2], f32(0))

    def row(r, v):                                                  # mul(M, v): row r dot v
        return ((m[4 * r] * v[0] + m[4 * r + 1] * v[1]) + m[4 * r + 2] * v[2]) + m[4 * r + 3] * v[3]
    origin = tuple(row(r, o4) for r in range(3))
    dw = tuple(row(r, d4) for r in range(3))
    length = np.sqrt(_dot(dw, dw))                                  # normalize
    direction = tuple(x / length for x in dw)
    with np.errstate(divide="ignore"):
        inv = tuple(f32(1) / x for x in direction)
    return origin, direction, inv


class Scene:
    """the six buffers bound at Sc/RaytracingMeshDrawer.cs:65-70 as numpy arrays"""

    def __init__(self, sorted_indices, tri_min, tri_max, internal, leaf, bvh_min, bvh_max, a, b, c):
        self.sorted_indices, self.tri_min, self.tri_max = sorted_indices, tri_min, tri_max
        self.internal, self.leaf, self.bvh_min, self.bvh_max = internal, leaf, bvh_min, bvh_max
        self.a, self.b, self.c = a, b, c


def _check_triangle(s, tri, ray, result, counters, fast_rule=False):  # :89-103
    origin, direction, inv = ray
    counters[2] += 1
    if ray_box_intersection(s.tri_min[tri], s.tri_max[tri], origin, inv):
        counters[3] += 1
        dist, u, v = ray_triangle_intersection(origin, direction, tuple(s.a[tri]), tuple(s.b[tri]), tuple(s.c[tri]))
        if fast_rule and dist < ray_box_entry(s.tri_min[tri], s.tri_max[tri], origin, inv):
            return result                                            # NOT the reference: the fast modes' accept rule
        if dist < result[0]:
            return [dist, tri, u, v]
    return result


def raytracing_thread(s, cam, idx, idy, counters, fast_rule=False):
    """kernel Raytracing for dispatch thread (idx, idy) up to :176.  Returns [distance, triangleIndex, u, v];
    counters = [pops, boxes hit, leaf AABB tests, triangle tests] (SURVEY 8d's P, B, L, T)."""
    ray = make_ray(cam, idx, idy)
    origin, direction, inv = ray
    result = [MAX_FLOAT, 0, f32(0), f32(0)]                         # :128-131
    stack = [0] * 64                                                # :133
    current = 0
    stack[current] = 0
    current = 1
    while current != 0:                                             # :138
        current -= 1
        index = stack[current]
        counters[0] += 1
        if not ray_box_intersection(s.bvh_min[index], s.bvh_max[index], origin, inv):
            continue
        counters[1] += 1
        left_index, left_type = int(s.internal[index][0]), int(s.internal[index][1])
        if left_type == INTERNAL_NODE:
            stack[current] = left_index
            current += 1
        else:
            tri = int(s.sorted_indices[int(s.leaf[left_index][1])])                  # :158
            result = _check_triangle(s, tri, ray, result, counters, fast_rule)
        right_index, right_type = int(s.internal[index][2]), int(s.internal[index][3])
        if right_type == INTERNAL_NODE:
            stack[current] = right_index
            current += 1
        else:
            tri = int(s.sorted_indices[int(s.leaf[right_index][1])])                 # :173
            result = _check_triangle(s, tri, ray, result, counters, fast_rule)
    return result


def raytracing(s, cam, x_step=1, y_step=1):
    """every (x, y) of the sampled grid; returns t, tri, u, v arrays [h, w] and the summed counters + hit count"""
    w = (cam["screen_width"] + x_step - 1) // x_step
    h = (cam["screen_height"] + y_step - 1) // y_step
    t = np.zeros((h, w), dtype=np.float32)
    tri = np.zeros((h, w), dtype=np.uint32)
    u = np.zeros((h, w), dtype=np.float32)
    v = np.zeros((h, w), dtype=np.float32)
    counters = [0, 0, 0, 0]
    for j in range(h):
        for i in range(w):
            r = raytracing_thread(s, cam, i * x_step, j * y_step, counters)
            t[j, i], tri[j, i], u[j, i], v[j, i] = r
    hits = int((t != MAX_FLOAT).sum())                              # :184 alpha
    return t, tri, u, v, counters + [hits]


# ---------------------------------------------------------------------------------------------------------------
# Sh/Raytracing/Raytracing.compute:178-184 — shading of one RaycastResult
# ---------------------------------------------------------------------------------------------------------------
def _sample_bilinear_clamp(tex, uu, vv):
    """SampleLevel(linearClampSampler, uv, 0) as DESIGN.md defines it: texel centres at (i + 0.5) / size, fp32 weights,
    clamp addressing, RGBA8 / 255 (the D3D sampler's fixed-point weights are not reproducible; stated there)."""
    hgt, wid = tex.shape[0], tex.shape[1]
    x = f32(uu) * f32(wid) - f32(0.5)
    y = f32(vv) * f32(hgt) - f32(0.5)
    xf, yf = np.floor(x), np.floor(y)
    fx, fy = x - xf, y - yf

    def clamp(q, hi):
        return int(min(max(q, f32(0)), f32(hi)))
    x0, x1, y0, y1 = clamp(xf, wid - 1), clamp(xf + f32(1), wid - 1), clamp(yf, hgt - 1), clamp(yf + f32(1), hgt - 1)
    gx, gy = f32(1) - fx, f32(1) - fy
    out = []
    for k in range(4):
        c00, c10 = f32(tex[y0, x0, k]) / f32(255), f32(tex[y0, x1, k]) / f32(255)
        c01, c11 = f32(tex[y1, x0, k]) / f32(255), f32(tex[y1, x1, k]) / f32(255)
        out.append((c00 * gx + c10 * fx) * gy + (c01 * gx + c11 * fx) * fy)
    return out


def shade(result, tri_record, tex):
    """result = [distance, triangleIndex, u, v]; tri_record = dict with a_uv .. c_normal of triangleData[triangleIndex];
    returns 4 float16 (the RGBA16F store of _outputTexture)."""
    uu, vv = f32(result[2]), f32(result[3])
    w = (f32(1) - uu) - vv
    tu = (w * f32(tri_record["a_uv"][0]) + uu * f32(tri_record["b_uv"][0])) + vv * f32(tri_record["c_uv"][0])
    tv = (w * f32(tri_record["a_uv"][1]) + uu * f32(tri_record["b_uv"][1])) + vv * f32(tri_record["c_uv"][1])
    nrm = [(w * f32(tri_record["a_normal"][k]) + uu * f32(tri_record["b_normal"][k])) + vv * f32(tri_record["c_normal"][k])
           for k in range(3)]
    light = f32(0.57735026)                           # `const float lightDir = normalize(float3(1,1,1))`: the scalar .x
    lam = _fmax(f32(0.4), (light * nrm[0] + light * nrm[1]) + light * nrm[2])
    col = _sample_bilinear_clamp(tex, tu, tv)
    rgba = [col[0] * lam, col[1] * lam, col[2] * lam, f32(1.0) if result[0] != MAX_FLOAT else f32(0.0)]
    return np.array(rgba, dtype=np.float32).astype(np.float16)


# ---------------------------------------------------------------------------------------------------------------
# the whole Awake() chain (Sc/RaytracingMeshDrawer.cs:30-51) on vertex arrays
# ---------------------------------------------------------------------------------------------------------------
def awake(a, b, c, capacity=None):
    """Returns a dict of every array the reference holds after Awake(): morton (unsorted), sorted_keys (after
    DistributeKeys), sorted_indices, tri_min / tri_max, internal, leaf, bvh_min / bvh_max, and the Scene."""
    a, b, c = (np.ascontiguousarray(x, dtype=np.float32) for x in (a, b, c))
    n = len(a)
    cap = n if capacity is None else int(capacity)
    morton, tri_min, tri_max = morton_aabb(a, b, c)
    keys = np.full(cap, U32, dtype=np.uint32)         # DataBuffer<uint>(.., uint.MaxValue)  :108-109
    idx = np.full(cap, U32, dtype=np.uint32)
    keys[:n] = morton
    idx[:n] = np.arange(n, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")           # ComputeBufferSorter.Sort(): stable LSD radix over the capacity
    keys, idx = keys[order], idx[order]
    keys = np.concatenate([distribute_keys(keys[:n], n), keys[n:]])
    internal, leaf = tree_constructor(keys, n, cap)
    bvh_min, bvh_max = bvh_constructor(n, internal, leaf, idx, tri_min, tri_max)
    scene = Scene(idx, tri_min, tri_max, internal, leaf, bvh_min, bvh_max, a, b, c)
    return {"n": n, "capacity": cap, "morton": morton, "sorted_keys": keys, "sorted_indices": idx, "tri_min": tri_min,
            "tri_max": tri_max, "internal": internal, "leaf": leaf, "bvh_min": bvh_min, "bvh_max": bvh_max, "scene": scene}
