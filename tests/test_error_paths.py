"""Error paths on the GPU: LBVH_ERR_OUT_OF_MEMORY from lbvh_buffer_alloc and from a failed growth of
context-owned scratch in the middle of a sort / a build (fault injection: lbvh_debug_switch LBVH_DEBUG_FAIL_RESERVE — on a healthy
288 GB device hipMalloc does not fail on its own).  After every failure the SAME context must still sort and build, bit-exact:
lbvh_reserve frees the old block before it allocates (lbvh_api.hip), so a failure must leave the slot empty, never dangling.
Reference behaviour being mirrored: none — Unity logs and carries on (SURVEY 8b "Error convention"); the build returns a status."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O                                                   # noqa: E402
from unitysimpleraytracing_amd import layouts as L, scenes           # noqa: E402

pytestmark = pytest.mark.gpu
ERR_OUT_OF_MEMORY = -2


def N():
    from unitysimpleraytracing_amd import _native
    return _native


def H():
    from unitysimpleraytracing_amd import host
    return host


def words(a):
    return np.ascontiguousarray(a).view(np.uint32)


def sort_once(c, keys, vals):
    kb, vb = H().DataBuffer(c, len(keys), np.uint32), H().DataBuffer(c, len(keys), np.uint32)
    kb.local[:] = keys; vb.local[:] = vals; kb.sync(); vb.sync()
    rc = N().lib.lbvh_sort_pairs(c.handle, kb.device, vb.device, len(keys))
    out = (kb.get_data().copy(), vb.get_data().copy()) if rc == 0 else None
    kb.dispose(); vb.dispose()
    return rc, out


def test_the_status_codes_are_the_headers():
    import re
    src = open(os.path.join(ROOT, "include", "lbvh.h")).read()
    m = re.search(r"#define\s+LBVH_ERR_OUT_OF_MEMORY\s+(-?\d+)", src)
    assert m and int(m.group(1)) == ERR_OUT_OF_MEMORY


def test_a_buffer_the_device_cannot_hold_is_out_of_memory_and_the_context_lives_on():
    c = H().Context(0)
    try:
        p = C.c_void_p()
        rc = N().lib.lbvh_buffer_alloc(c.handle, C.c_size_t(1 << 44), C.c_size_t(1), C.byref(p))        # 16 TiB
        assert rc == ERR_OUT_OF_MEMORY and not p.value
        assert b"hipMalloc" in N().lib.lbvh_last_error(c.handle)
        rc = N().lib.lbvh_buffer_alloc(c.handle, C.c_size_t(1 << 44), C.c_size_t(1 << 30), C.byref(p))  # count * stride wraps
        assert rc == ERR_OUT_OF_MEMORY and not p.value
        # the context still sorts and builds
        rng = np.random.default_rng(3)
        keys = rng.integers(0, 1 << 32, 200001, dtype=np.uint64).astype(np.uint32)
        vals = rng.permutation(len(keys)).astype(np.uint32)
        rc, out = sort_once(c, keys, vals)
        ok, ov = O.sort_pairs(keys, vals)
        assert rc == 0 and (out[0] == ok).all() and (out[1] == ov).all()
        tris = scenes.random_triangles(30000, seed=5)
        d = H().RaytracingMeshDrawer(c, tris).awake()
        b = O.Built(tris, capacity=d.container.capacity, threads=8)
        d.container.get_all_gpu_data()
        assert (words(d.container.bvh_internal_node.local) == words(b.internal)).all()
        d.on_destroy()
    finally:
        c.close()


def test_a_failed_scratch_growth_inside_the_sort_leaves_a_usable_context():
    c = H().Context(0)
    try:
        rng = np.random.default_rng(4)
        small = rng.integers(0, 1 << 32, 50000, dtype=np.uint64).astype(np.uint32)
        big = rng.integers(0, 1 << 32, 3000001, dtype=np.uint64).astype(np.uint32)
        vs, vb = rng.permutation(len(small)).astype(np.uint32), rng.permutation(len(big)).astype(np.uint32)
        rc, out = sort_once(c, small, vs)                                  # scratch sized for 50 000 pairs
        assert rc == 0
        c.debug_switch(N().DEBUG_SWITCH_FAIL_RESERVE, 1)                   # the growth for 3 M pairs fails
        rc, _ = sort_once(c, big, vb)
        assert rc == ERR_OUT_OF_MEMORY
        assert b"LBVH_DEBUG_FAIL_RESERVE" in N().lib.lbvh_last_error(c.handle)
        for keys, vals in ((small, vs), (big, vb), (small, vs)):           # the slot was left empty: both sizes sort again
            rc, out = sort_once(c, keys, vals)
            ok, ov = O.sort_pairs(keys, vals)
            assert rc == 0 and (out[0] == ok).all() and (out[1] == ov).all()
    finally:
        c.close()


@pytest.mark.parametrize("kth", [1, 2, 3, 4, 5, 6])
def test_a_failed_scratch_growth_inside_the_build_leaves_a_usable_context(kth):
    """the k-th scratch growth of a first lbvh_build_scene fails (sort scratch, derived-scene keys, range hierarchy, scan scratch ...:
    whichever comes k-th); the call reports it, and the next call on the same context builds the scene bit-exact"""
    c = H().Context(0)
    try:
        tris = scenes.tiled_torus(nu=30, nv=20, grid=3)
        d = H().RaytracingMeshDrawer(c, tris)
        c.debug_switch(N().DEBUG_SWITCH_FAIL_RESERVE, kth)
        failed = False
        try:
            d.awake()
            d.rebuild()
        except N().LbvhError as e:
            failed = True
            assert e.status == ERR_OUT_OF_MEMORY, (e.status, str(e))
        c.debug_switch(N().DEBUG_SWITCH_FAIL_RESERVE, 0)
        if not failed:
            pytest.skip(f"fewer than {kth} scratch growths in awake() + rebuild()")
        c.sync()
        d2 = H().RaytracingMeshDrawer(c, tris).awake()
        for _ in range(2):
            d2.rebuild()
        cont = d2.container
        b = O.Built(tris, capacity=cont.capacity, threads=8)
        bad_leaf, bad_inner = cont.get_all_gpu_data()
        assert len(bad_leaf) == 0 and len(bad_inner) == 0
        n = b.n
        assert (cont.keys.local == b.keys).all() and (cont.triangle_index.local == b.indices).all()
        assert (words(cont.bvh_internal_node.local) == words(b.internal)).all() and (words(cont.bvh_leaf_node.local) == words(b.leaf)).all()
        assert (cont.bvh_data.local["min"][: n - 1] == b.bvh["min"][: n - 1]).all() and (cont.bvh_data.local["max"][: n - 1] == b.bvh["max"][: n - 1]).all()
        cam = scenes.camera(160, 90, (0.0, 0.0, 250.0))
        d2.update(cam, mode=L.TRACE_REFERENCE)
        ref = d2.hits()
        d2.update(cam, mode=L.TRACE_FAST)
        assert (d2.hits()["t"] == ref["t"]).all()
        d2.on_destroy()
    finally:
        c.close()
