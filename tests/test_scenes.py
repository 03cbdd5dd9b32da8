"""Host-side scene helpers (CPU)."""
import numpy as np

from unitysimpleraytracing_amd import layouts as L
from unitysimpleraytracing_amd import scenes

OBJ = """# two quads and a triangle, mixed corner formats, negative indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 2 0 0
v 2 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
f 1/1/1 2/2/1 3/3/1 4/4/1
f 2/1/1 5/2/1 6/3/1 3/4/1
f -6/1/1 -5/2/1 -4/3/1
"""


def test_load_obj_fans_polygons_and_keeps_attributes():
    t = scenes.load_obj(OBJ, is_text=True)
    assert t.dtype == L.TRIANGLE and len(t) == 5
    assert t["a"][0].tolist() == [0, 0, 0] and t["b"][0].tolist() == [1, 0, 0] and t["c"][0].tolist() == [1, 1, 0]
    assert t["a"][1].tolist() == [0, 0, 0] and t["b"][1].tolist() == [1, 1, 0] and t["c"][1].tolist() == [0, 1, 0]   # a c d
    assert t["b"][2].tolist() == [2, 0, 0]
    assert t["a"][4].tolist() == [0, 0, 0] and t["c"][4].tolist() == [1, 1, 0]                # negative indices
    assert t["c_uv"][0].tolist() == [1, 1] and (t["a_normal"] == (0, 0, 1)).all()


def test_load_obj_without_uv_and_normals_uses_face_normals():
    t = scenes.load_obj("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", is_text=True)
    assert len(t) == 1 and t["a_normal"][0].tolist() == [0, 0, 1] and t["b_uv"][0].tolist() == [0, 0]


def test_grid_scene_matches_the_reference_asset_shape():
    g = scenes.grid_scene()
    assert len(g) == 12800                                    # 6 400 quads (Assets/_Assets/ExampleObject3.obj)
    assert np.allclose(g["a"].min(axis=0), (-4, -4, 0)) and np.allclose(g["c"].max(axis=0), (4, 4, 0))
    assert scenes.capacity_for(12800) == 13312 and scenes.capacity_for(1024) == 1024


def test_tiled_torus_is_deterministic_and_inside_the_morton_box():
    a = scenes.tiled_torus(nu=8, nv=6, grid=2)
    b = scenes.tiled_torus(nu=8, nv=6, grid=2)
    assert len(a) == 2 * 8 * 6 * 8 and (a == b).all()
    for f in ("a", "b", "c"):
        assert np.abs(a[f]).max() < 125.0


def test_reference_assets_through_load_obj():
    """the reference's own OBJ assets through the ingest.  /root/reference exists only in the build
    container; the triangles it yields there are what tests/golden/{example_object3,viking_room}.npz hold (those files
    carry the data to the GPU box, where the same triangles go through build -> trace -> shade)."""
    import os
    import numpy as np
    from unitysimpleraytracing_amd import layouts as L
    assets = "/root/reference/Assets/_Assets"
    if not os.path.isdir(assets):
        import pytest
        pytest.skip("no /root/reference on this box")
    golden = os.path.join(os.path.dirname(__file__), "golden")
    for obj, fixture, faces in (("ExampleObject3.obj", "example_object3.npz", 12800), ("viking_room.obj", "viking_room.npz", 3828)):
        t = scenes.load_obj(os.path.join(assets, obj))
        assert t.dtype == L.TRIANGLE and len(t) == faces
        g = np.load(os.path.join(golden, fixture))["triangles"]
        assert t.tobytes() == np.ascontiguousarray(g, dtype=L.TRIANGLE).tobytes()
    # the quad grid: 6 400 quads fanned into 12 800 triangles, all on z = 0 inside [-4, 4]^2, uv and normals carried
    t = scenes.load_obj(os.path.join(assets, "ExampleObject3.obj"))
    assert (t["a"][:, 2] == 0).all() and abs(t["a"][:, :2]).max() <= 4.0
    assert (t["a_normal"][:, 2] != 0).all() and t["a_uv"].min() >= 0.0 and t["c_uv"].max() <= 1.0


# ---- the compiled host's ingest (host/lbvh_mesh.hpp) against the Python twin -------------------------------------------

def _cpp_triangles(obj_text, tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "unitysimpleraytracing_amd", "host", "obj_to_triangles")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    src, out = tmp_path / "mesh.obj", tmp_path / "mesh.bin"
    src.write_text(obj_text)
    r = subprocess.run([exe, str(src), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return np.fromfile(out, dtype=L.TRIANGLE)


def test_cpp_obj_ingest_equals_the_python_ingest(tmp_path):
    """lbvh::LoadObj + lbvh::MeshTriangles (what `new MeshBufferContainer(mesh)` gathers,
    MeshBufferContainer.cs:117-146) produce the 128-byte records scenes.load_obj does, byte for byte: polygons fanned,
    mixed corner formats, negative indices, comments, missing uv / normals (face normals in fp32), odd number formats."""
    rng = np.random.default_rng(7)
    pts = rng.uniform(-50, 50, (40, 3))
    soup = "".join(f"v {x:.9g} {y:.7e} {float(z)!r}\n" for x, y, z in pts)
    soup += "".join(f"f {a} {b} {c} {d}  # quad\n" for a, b, c, d in rng.integers(1, 41, (30, 4)))
    soup += "f -1 -2 -3\nf 1 1 1\n"                                    # relative indices; a zero-area face (normal 0)
    cases = [OBJ, "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", soup,
             "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 2//1 4//1 3//1\n",          # normals, no uv
             "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0.25 0.5\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\n"]                       # uv, no normals
    for text in cases:
        want = scenes.load_obj(text, is_text=True)
        got = _cpp_triangles(text, tmp_path)
        assert len(got) == len(want) and got.tobytes() == want.tobytes()


def test_cpp_obj_ingest_rejects_malformed_files(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "unitysimpleraytracing_amd", "host", "obj_to_triangles")
    for text in ("v 0 0\n", "v 0 0 0\nf 1 2 3\n", "v 0 0 x\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 0\n"):
        src = tmp_path / "bad.obj"
        src.write_text(text)
        r = subprocess.run([exe, str(src), str(tmp_path / "bad.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "OBJ line" in r.stderr, (text, r.stderr)
    assert subprocess.run([exe, str(tmp_path / "missing.obj"), str(tmp_path / "x.bin")], capture_output=True).returncode == 1


def test_cpp_obj_ingest_on_the_reference_assets(tmp_path):
    import glob
    import os
    objs = sorted(glob.glob("/root/reference/Assets/_Assets/*.obj"))
    if not objs:
        import pytest
        pytest.skip("no /root/reference on this box")
    for obj in objs:
        want = scenes.load_obj(obj)
        got = _cpp_triangles(open(obj).read(), tmp_path)
        assert len(want) > 0 and got.tobytes() == want.tobytes(), obj
