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
