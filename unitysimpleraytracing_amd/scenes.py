"""Synthetic scenes and cameras (numpy only; deterministic).

The reference ships no benchmark scene of the sizes BASELINE.json names (its capacity is hard-wired
to 524 288 slots, Assets/_Scripts/Constants.cs:3-6) and no bunny asset exists, so the configs are
synthesised here.
"""
import math

import numpy as np

from .layouts import TRIANGLE


def _pack_triangles(a, b, c, uv=None, normals=None):
    """a, b, c: (n, 3) float32 vertex positions -> TRIANGLE[n] (128-B reference layout)."""
    n = a.shape[0]
    t = np.zeros(n, dtype=TRIANGLE)
    t["a"], t["b"], t["c"] = a, b, c
    if uv is None:
        t["a_uv"] = (0.0, 0.0)
        t["b_uv"] = (1.0, 0.0)
        t["c_uv"] = (0.0, 1.0)
    else:
        t["a_uv"], t["b_uv"], t["c_uv"] = uv
    if normals is None:
        fn = np.cross(b - a, c - a).astype(np.float32)
        ln = np.linalg.norm(fn, axis=1, keepdims=True).astype(np.float32)
        ln[ln == 0] = 1.0
        fn = (fn / ln).astype(np.float32)
        t["a_normal"] = t["b_normal"] = t["c_normal"] = fn
    else:
        t["a_normal"], t["b_normal"], t["c_normal"] = normals
    return t


def random_triangles(n=4096, seed=1, extent=100.0, edge=2.0):
    """cfg1: n triangles, centre uniform in [-extent, extent]^3, two edge vectors uniform in
    [-edge, edge]^3."""
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-extent, extent, size=(n, 3)).astype(np.float32)
    e1 = rng.uniform(-edge, edge, size=(n, 3)).astype(np.float32)
    e2 = rng.uniform(-edge, edge, size=(n, 3)).astype(np.float32)
    a = centre
    b = (centre + e1).astype(np.float32)
    c = (centre + e2).astype(np.float32)
    return _pack_triangles(a, b, c)


def grid_scene(quads=80, half=4.0):
    """The reference's default mesh re-created procedurally: an 80x80 quad grid on z = 0,
    x, y in [-4, 4] (Assets/_Assets/ExampleObject3.obj, wired in Assets/__Scenes/Scene.unity:364),
    two triangles per quad = 12 800 triangles, normals +z, uv = (x, y) mapped to [0, 1]."""
    xs = np.linspace(-half, half, quads + 1, dtype=np.float32)
    gx, gy = np.meshgrid(xs, xs, indexing="xy")
    def vert(ix, iy):
        x = gx[iy, ix].ravel()
        y = gy[iy, ix].ravel()
        return np.stack([x, y, np.zeros_like(x)], axis=1).astype(np.float32)
    ix, iy = np.meshgrid(np.arange(quads), np.arange(quads), indexing="xy")
    v00, v10 = vert(ix, iy), vert(ix + 1, iy)
    v11, v01 = vert(ix + 1, iy + 1), vert(ix, iy + 1)
    a = np.concatenate([v00, v00])
    b = np.concatenate([v10, v11])
    c = np.concatenate([v11, v01])
    # interleave so the two triangles of a quad are adjacent (mesh order)
    order = np.arange(2 * quads * quads).reshape(2, -1).T.ravel()
    a, b, c = a[order], b[order], c[order]
    def uv_of(p):
        return ((p[:, :2] + half) / (2 * half)).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], dtype=np.float32), (a.shape[0], 1))
    return _pack_triangles(a, b, c, uv=(uv_of(a), uv_of(b), uv_of(c)), normals=(nrm, nrm, nrm))


def _bumpy_torus(nu, nv, major=12.0, minor=5.0):
    """Closed, non-convex blob: torus with a low-order sinusoidal bump on the minor radius,
    sampled on an nu x nv (u, v) quad grid, 2 triangles per quad.  Returns (a, b, c) float64."""
    u = np.arange(nu + 1) * (2 * math.pi / nu)
    v = np.arange(nv + 1) * (2 * math.pi / nv)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    r = minor * (1.0 + 0.25 * np.sin(3 * uu) * np.cos(2 * vv) + 0.1 * np.sin(5 * vv + uu))
    x = (major + r * np.cos(vv)) * np.cos(uu)
    y = (major + r * np.cos(vv)) * np.sin(uu)
    z = r * np.sin(vv)
    p = np.stack([x, y, z], axis=-1)
    p00 = p[:-1, :-1].reshape(-1, 3)
    p10 = p[1:, :-1].reshape(-1, 3)
    p11 = p[1:, 1:].reshape(-1, 3)
    p01 = p[:-1, 1:].reshape(-1, 3)
    a = np.concatenate([p00, p00])
    b = np.concatenate([p10, p11])
    c = np.concatenate([p11, p01])
    return a, b, c


def _rotation(rng):
    """Seeded random rotation matrix (QR of a Gaussian matrix, det fixed to +1)."""
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def tiled_torus(nu=80, nv=50, grid=5, pitch=45.0, seed=2, shuffle=True, with_bodies=False):
    """cfg2/3/5 ("tiled bunny" stand-in): one bumpy torus of 2*nu*nv faces tiled on a grid^3
    lattice with pitch `pitch`, centred in the reference's Morton box [-125, 125]^3, each tile with
    its own seeded rotation, triangle order shuffled.  Defaults: 8 000 faces x 125 tiles =
    1 000 000 triangles.  nu=400, nv=160 gives cfg4's 16 M."""
    rng = np.random.default_rng(seed)
    a0, b0, c0 = _bumpy_torus(nu, nv)
    faces = a0.shape[0]
    n = faces * grid ** 3
    a = np.empty((n, 3), dtype=np.float32)
    b = np.empty((n, 3), dtype=np.float32)
    c = np.empty((n, 3), dtype=np.float32)
    k = 0
    offs = (np.arange(grid) - (grid - 1) / 2.0) * pitch
    for ix in range(grid):
        for iy in range(grid):
            for iz in range(grid):
                rot = _rotation(rng)
                t = np.array([offs[ix], offs[iy], offs[iz]])
                sl = slice(k * faces, (k + 1) * faces)
                a[sl] = (a0 @ rot.T + t).astype(np.float32)
                b[sl] = (b0 @ rot.T + t).astype(np.float32)
                c[sl] = (c0 @ rot.T + t).astype(np.float32)
                k += 1
    body = np.repeat(np.arange(grid ** 3, dtype=np.uint32), faces)
    if shuffle:
        perm = rng.permutation(n)
        a, b, c = a[perm], b[perm], c[perm]
        body = body[perm]
    tris = _pack_triangles(a, b, c)
    if not with_bodies:
        return tris
    # cfg5 (dynamic scene): every tile is a rigid body rotating about the Y axis through its centre
    centres = np.zeros((grid ** 3, 4), dtype=np.float32)
    k = 0
    for ix in range(grid):
        for iy in range(grid):
            for iz in range(grid):
                centres[k, :3] = (offs[ix], offs[iy], offs[iz])
                k += 1
    return tris, body, centres


def load_obj(path_or_text, is_text=False):
    """Mesh ingest for Wavefront OBJ (the reference's assets are OBJ, Assets/_Assets/*.obj, imported by Unity and
    flattened to Triangle[] by Assets/_Scripts/MeshBufferContainer.cs:117-146): v / vt / vn, faces with
    v, v/vt, v//vn or v/vt/vn corners, negative indices, polygons fanned into triangles (a quad a b c d ->
    a b c, a c d).  Missing uv -> (0,0); missing normals -> the face normal.  Returns TRIANGLE[n]."""
    text = path_or_text if is_text else open(path_or_text).read()
    v, vt, vn = [], [], []
    corners = []          # per triangle: 3 x (vi, ti, ni), -1 = absent
    for line in text.splitlines():
        p = line.split("#", 1)[0].split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "vt":
            vt.append([float(x) for x in (p[1:3] + ["0"])[:2]])
        elif p[0] == "vn":
            vn.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            poly = []
            for c in p[1:]:
                idx = (c.split("/") + ["", ""])[:3]
                def rel(tok, count):
                    if tok == "":
                        return -1
                    k = int(tok)
                    return k - 1 if k > 0 else count + k
                poly.append((rel(idx[0], len(v)), rel(idx[1], len(vt)), rel(idx[2], len(vn))))
            for k in range(1, len(poly) - 1):
                corners.append((poly[0], poly[k], poly[k + 1]))
    n = len(corners)
    va = np.asarray(v, dtype=np.float32).reshape(-1, 3)
    c = np.asarray(corners, dtype=np.int64).reshape(n, 3, 3)
    a, b, cc = va[c[:, 0, 0]], va[c[:, 1, 0]], va[c[:, 2, 0]]
    uv = normals = None
    if len(vt) and (c[:, :, 1] >= 0).all():
        ta = np.asarray(vt, dtype=np.float32)
        uv = (ta[c[:, 0, 1]], ta[c[:, 1, 1]], ta[c[:, 2, 1]])
    if len(vn) and (c[:, :, 2] >= 0).all():
        na = np.asarray(vn, dtype=np.float32)
        normals = (na[c[:, 0, 2]], na[c[:, 1, 2]], na[c[:, 2, 2]])
    t = _pack_triangles(a, b, cc, uv=uv, normals=normals)
    if uv is None:
        t["a_uv"] = t["b_uv"] = t["c_uv"] = (0.0, 0.0)
    return t


def camera(width, height, position, fov_y_deg=60.0, near=0.3):
    """Camera uniforms as RaytracingMeshDrawer.Update sets them
    (Assets/_Scripts/RaytracingMeshDrawer.cs:78-81) for a Unity camera at `position` rotated 180
    degrees about Y (quaternion (0,1,0,0), Assets/__Scenes/Scene.unity:342-343), i.e. looking
    toward -Z in world space.  cameraToWorldMatrix = TRS * diag(1,1,-1,1): rows
    [-1,0,0,px], [0,1,0,py], [0,0,1,pz], [0,0,0,1].  Returns a dict of plain values."""
    px, py, pz = (float(v) for v in position)
    m = np.array([[-1, 0, 0, px], [0, 1, 0, py], [0, 0, 1, pz], [0, 0, 0, 1]], dtype=np.float32)
    fov = np.float32(math.tan(math.radians(fov_y_deg) / 2.0))   # Mathf.Tan(fov * Deg2Rad / 2)
    return {
        "screen_width": int(width),
        "screen_height": int(height),
        "camera_fov": float(fov),
        "near_plane": float(np.float32(near)),
        "camera_to_world": m.reshape(-1).copy(),
    }


def reference_scene_camera(width=1024, height=1024):
    """The camera of the reference scene: position (0, 0, 15.7), fov 60, near 0.3
    (Assets/__Scenes/Scene.unity:315-317, 342-343); default 1024x1024 window
    (ProjectSettings/ProjectSettings.asset:45-46)."""
    return camera(width, height, (0.0, 0.0, 15.7))


def capacity_for(n, tile=1024):
    """Padded capacity: the reference sorts whole THREADS_PER_BLOCK(1024)-key tiles
    (Assets/_Scripts/Constants.cs:3-6 fixes 512 of them); here the smallest multiple >= n."""
    return ((int(n) + tile - 1) // tile) * tile
