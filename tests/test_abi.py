"""The C-ABI library loads on a CPU-only box and exports every symbol include/lbvh.h (the drop-in boundary) and
include/lbvh_debug.h (test hooks and measurement aids) declare.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="lbvh.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
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
    for name in declared_functions() + declared_functions("lbvh_debug.h"):
        assert hasattr(N.lib, name), f"liblbvh.so does not export {name}"
    assert set(N.SIGNATURES) == set(declared_functions())
    assert set(N.DEBUG_SIGNATURES) == set(declared_functions("lbvh_debug.h"))
    assert N.lib.lbvh_abi_version() == N.ABI_VERSION == 11


def test_the_boundary_header_holds_no_test_hooks():
    """what a Unity maintainer reads (include/lbvh.h, LbvhNative.cs) is the product's calls; hooks, probes and
    per-kernel profiling live in lbvh_debug.h / LbvhNativeDebug.cs, and the library reads no environment variable."""
    names = declared_functions()
    assert not [f for f in names if "debug" in f or "probe" in f or "profile" in f or f in ("lbvh_ray_stats_target", "lbvh_trace_tile_costs")]
    assert set(names).isdisjoint(declared_functions("lbvh_debug.h"))
    src = os.path.join(ROOT, "unitysimpleraytracing_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(src, f)).read(), f
    # none of the re-hosted reference classes needs the debug table
    cs = os.path.join(ROOT, "bindings", "csharp")
    for f in os.listdir(cs):
        if f.endswith(".Native.cs") or f in ("NativeBuffer.cs", "LbvhContext.cs"):
            assert "LbvhNativeDebug" not in open(os.path.join(cs, f)).read(), f


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


@pytest.mark.parametrize("queues,group,tiles", [(1, 1, 1), (1, 16, 1000), (8, 1, 7), (8, 8, 130), (8, 8, 1000), (8, 16, 8192),
                                                (8, 16, 2049)])
def test_sort_ticket_order_model(queues, group, tiles):
    """CPU model of the sort's tile hand-out (lbvh_sort.hip): the decoupled look-back of tile T spins on
    every lower tile, so a lower tile must never be stranded behind workgroups that wait for it.
      * the ticket -> tile map is a bijection onto a superset of [0, tiles): every tile is handed out exactly once;
      * one queue (what every context but the full 8-XCD device uses): tiles are handed out in increasing order, so
        whatever order workgroups start in, everything a tile waits for is already running;
      * eight queues, workgroups dealt round-robin over the XCDs (the layout lbvh_create selects them for: workgroup
        b belongs to XCD b % 8, every XCD starts its own workgroups as its slots free up): with any number R of
        resident workgroups per XCD, the lowest tile not yet handed out is always the next ticket of an XCD whose
        residents can all finish — simulated with the worst case where a tile finishes only when every lower tile
        has been handed out."""
    from unitysimpleraytracing_amd import _native as N
    f = N.lib.lbvh_debug_sort_ticket_tile
    per_queue = []
    for x in range(queues):
        ks, k = [], 0
        while True:
            t = f(k, x, group, queues)
            if t >= tiles and k % group == 0:
                break
            ks.append(t)
            k += 1
        assert ks == sorted(ks)
        per_queue.append([t for t in ks if t < tiles])
    everything = sorted(t for q in per_queue for t in q)
    assert everything == list(range(tiles))                              # bijection
    if queues == 1:
        assert per_queue[0] == list(range(tiles))
        return
    # round-robin placement: workgroup b runs on XCD b % 8 and takes its home queue first, then the others in turn
    for resident in (1, 3):
        nxt = [0] * queues                                               # next ticket per queue
        handed = set()
        running = [[] for _ in range(queues)]                            # tiles resident per XCD
        quota = [len(range(x, tiles, queues)) for x in range(queues)]    # workgroups of the grid that belong to XCD x
        started = [0] * queues

        def take(home):
            for a in range(queues):
                x = (home + a) % queues
                while True:
                    t = f(nxt[x], x, group, queues)
                    nxt[x] += 1
                    if t < tiles:
                        return t
                    if nxt[x] > tiles + queues * group:                   # this queue is drained
                        break
            raise AssertionError("no tile left for a workgroup")

        while len(handed) < tiles or any(running):
            progressed = False
            for x in range(queues):                                      # fill free slots in dispatch order
                while len(running[x]) < resident and started[x] < quota[x]:
                    t = take(x)
                    handed.add(t)
                    running[x].append(t)
                    started[x] += 1
                    progressed = True
            # a tile can finish once every lower tile has been handed out (its look-back then completes)
            low = 0
            while low in handed:
                low += 1
            for x in range(queues):
                done = [t for t in running[x] if t < low]
                if done:
                    running[x] = [t for t in running[x] if t >= low]
                    progressed = True
            assert progressed, f"stranded: lowest missing tile {low}, resident {running}"


def test_csharp_binding_declares_every_entry_point_with_the_headers_arity():
    """bindings/csharp/LbvhNative.cs cannot be compiled here (no C# toolchain): at least every function of include/lbvh.h
    has a [DllImport] of the same name and parameter count, nothing else is imported, and each name is exported."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from unitysimpleraytracing_amd import _native as N
    for header, binding, least in (("lbvh.h", "LbvhNative.cs", 40), ("lbvh_debug.h", "LbvhNativeDebug.cs", 8)):
        hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", header)).read(), flags=re.S)
        protos = re.findall(r"\b(?:lbvh_status|int32_t|uint32_t|uint64_t|const char\*)\s+(lbvh_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
        c = {name: (0 if args.strip() in ("", "void") else len(args.split(","))) for name, args in protos}
        cs = re.sub(r"//.*", "", open(os.path.join(root, "bindings", "csharp", binding)).read())
        imports = re.findall(r"\[DllImport\(Lib\)\]\s*public static extern \w+\s+(lbvh_\w+)\s*\(([^;]*?)\)\s*;", cs, flags=re.S)
        d = {name: (0 if not args.strip() else len(args.split(","))) for name, args in imports}
        assert len(c) >= least and set(c) == set(d), (sorted(set(c) - set(d)), sorted(set(d) - set(c)))
        assert not [(k, c[k], d[k]) for k in c if c[k] != d[k]]
        for name in d:
            assert hasattr(N.lib, name), name
