"""The C-ABI library loads on a CPU-only box and exports every symbol include/lbvh.h declares.
No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "lbvh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbvh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("lbvh_create", "lbvh_destroy", "lbvh_buffer_alloc", "lbvh_buffer_upload", "lbvh_buffer_download",
                 "lbvh_morton_aabb", "lbvh_sort_pairs", "lbvh_distribute_keys", "lbvh_build_tree", "lbvh_refit",
                 "lbvh_trace_primary", "lbvh_build_fast_scene"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from unitysimpleraytracing_amd import _native as N
    for name in declared_functions():
        assert hasattr(N.lib, name), f"liblbvh.so does not export {name}"
    assert set(N.SIGNATURES) == set(declared_functions())
    assert N.lib.lbvh_abi_version() == N.ABI_VERSION == 4


def test_struct_layouts_match_the_reference():
    from unitysimpleraytracing_amd import _native as N
    from unitysimpleraytracing_amd import layouts as L
    assert L.TRIANGLE.itemsize == 128 and L.AABB.itemsize == 32          # MeshBufferContainer.cs:98-106
    assert L.INTERNAL_NODE.itemsize == 24 and L.LEAF_NODE.itemsize == 8
    assert [L.TRIANGLE.fields[f][1] for f in ("a", "b", "c", "a_uv", "b_uv", "c_uv", "a_normal", "b_normal",
                                               "c_normal")] == [0, 16, 32, 48, 56, 64, 80, 96, 112]
    assert [L.AABB.fields[f][1] for f in ("min", "max")] == [0, 16]
    assert C.sizeof(N.Camera) == 80 and C.sizeof(N.Scene) == 56


def test_errors_without_a_gpu_are_loud():
    """On a box with no GPU, context creation fails with a status and a message; nothing falls
    back to the CPU."""
    from unitysimpleraytracing_amd import _native as N
    if N.lib.lbvh_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert N.lib.lbvh_create(0, C.byref(h)) == -4
    assert b"no HIP device" in N.lib.lbvh_last_error(None)
    with pytest.raises(N.LbvhError):
        from unitysimpleraytracing_amd.host import Context
        Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "unitysimpleraytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "liblbvh_oracle" not in text, f
