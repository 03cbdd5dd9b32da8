"""Host side of the hot path: the reference's C# classes mirrored over the C ABI.

Same names, call order and argument meaning as Assets/_Scripts/{DataBuffer, MeshBufferContainer,
ComputeBufferSorter, BVHConstructor, RaytracingMeshDrawer}.cs, so a test reads like the
reference's Awake()/Update().  Every method is one C-ABI call (include/lbvh.h); nothing here
computes.  (The compiled-language twin of this file is host/lbvh_host.hpp; the C# [DllImport]
shim a Unity maintainer would add is in INTEGRATION.md.)
"""
import ctypes as C

import numpy as np

from . import _native as N
from . import layouts as L
from .scenes import capacity_for


class Context:
    """One GPU + one HIP stream.  Stands in for the implicit Unity graphics device and the
    IShaderContainer kernel registry (Assets/_Scripts/ShaderContainer.cs:6-40)."""

    def __init__(self, device_id=0, stream=None):
        h = C.c_void_p()
        if stream is None:
            N.check(None, N.lib.lbvh_create(device_id, C.byref(h)))
        else:
            N.check(None, N.lib.lbvh_create_on_stream(device_id, C.c_void_p(stream), C.byref(h)))
        self.handle = h
        self.device_id = device_id

    def sync(self):
        N.check(self.handle, N.lib.lbvh_sync(self.handle))

    def close(self):
        if self.handle:
            N.lib.lbvh_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- measurement helpers ------------------------------------------------------------------
    def event(self):
        e = C.c_void_p()
        N.check(self.handle, N.lib.lbvh_event_create(self.handle, C.byref(e)))
        return e

    def record(self, ev):
        N.check(self.handle, N.lib.lbvh_event_record(self.handle, ev))

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        N.check(self.handle, N.lib.lbvh_event_elapsed_ms(self.handle, start, stop, C.byref(ms)))
        return ms.value

    def destroy_event(self, ev):
        N.lib.lbvh_event_destroy(self.handle, ev)

    def profile_begin(self):
        N.check(self.handle, N.lib.lbvh_profile_begin(self.handle))

    def profile_end(self):
        """{kernel name: (launches, total device ms)} since profile_begin."""
        rows = (N.ProfileRow * 64)()
        n = C.c_int32()
        N.check(self.handle, N.lib.lbvh_profile_end(self.handle, rows, 64, C.byref(n)))
        return {rows[i].name.decode(): (int(rows[i].launches), float(rows[i].total_ms)) for i in range(n.value)}

    def clock_probe(self):
        """shader clock held under a vector-ALU-bound load, MHz (lbvh_clock_probe)"""
        mhz = C.c_float()
        N.check(self.handle, N.lib.lbvh_clock_probe(self.handle, C.byref(mhz)))
        return mhz.value

    def trace_forget(self):
        """drop the traversal's dispatch history: the next LBVH_TRACE_FAST frame is a cold one"""
        N.check(self.handle, N.lib.lbvh_trace_forget(self.handle))

    def trace_costs_export(self, frame_costs, tiles_x, tiles_y):
        """this context's per-tile step counts of its last LBVH_TRACE_FAST trace into a full-frame DataBuffer (u32 per tile)"""
        N.check(self.handle, N.lib.lbvh_trace_costs_export(self.handle, frame_costs.device, tiles_x, tiles_y))

    def trace_costs_import(self, frame_costs, tiles_x, tiles_y):
        """the merged per-tile step counts of every rank's last trace: the next moved-camera frame's dispatch hint"""
        N.check(self.handle, N.lib.lbvh_trace_costs_import(self.handle, frame_costs.device, tiles_x, tiles_y))

    def copy_probe(self, dst, src, nbytes):
        N.check(self.handle, N.lib.lbvh_copy_bandwidth_probe(self.handle, dst, src, nbytes))

    # -- one frame from N GPUs (include/lbvh.h, "one frame from N GPUs") ---------------------------------
    def peer_enable(self, peer_device):
        """kernels of this context may store into memory of `peer_device` (the frame's owner)"""
        N.check(self.handle, N.lib.lbvh_peer_enable(self.handle, int(peer_device)))

    def sync_event(self):
        """an ORDERING event (system-scope release at its record), for wait_event of another context"""
        e = C.c_void_p()
        N.check(self.handle, N.lib.lbvh_sync_event_create(self.handle, C.byref(e)))
        return e

    def wait_event(self, ev):
        """this context's stream waits on the device for an event recorded on any context of the process"""
        N.check(self.handle, N.lib.lbvh_event_wait(self.handle, ev))

    def ipc_export(self, device_ptr):
        """64-byte cross-process handle of a buffer of this context (hipIpcGetMemHandle)"""
        h = (C.c_uint8 * 64)()
        N.check(self.handle, N.lib.lbvh_ipc_export(self.handle, device_ptr, h))
        return bytes(h)

    def ipc_import(self, handle_bytes):
        """the exporting process's buffer mapped here: a device pointer usable on this context"""
        h = (C.c_uint8 * 64).from_buffer_copy(handle_bytes)
        p = C.c_void_p()
        N.check(self.handle, N.lib.lbvh_ipc_import(self.handle, h, C.byref(p)))
        return p

    def ipc_close(self, device_ptr):
        N.check(self.handle, N.lib.lbvh_ipc_close(self.handle, device_ptr))

    def flags_alloc(self, n_words):
        """n_words zeroed completion flags a running kernel may poll while another GPU / process stores into them (uncached
        device memory, lbvh_flags_alloc); free with flags_free"""
        p = C.c_void_p()
        N.check(self.handle, N.lib.lbvh_flags_alloc(self.handle, int(n_words), C.byref(p)))
        return p

    def flags_free(self, flags_ptr):
        N.check(self.handle, N.lib.lbvh_buffer_free(self.handle, flags_ptr))

    def debug_switch(self, which, value):
        """include/lbvh_debug.h: test / measurement switches of this context (all 0 in the product)"""
        N.check(self.handle, N.lib.lbvh_debug_switch(self.handle, int(which), int(value)))

    def frame_signal(self, flags_ptr, slot, value):
        """flags[slot] := value (system-scope release) once everything enqueued so far has finished"""
        N.check(self.handle, N.lib.lbvh_frame_signal(self.handle, flags_ptr, slot, value))

    def frame_wait(self, flags_ptr, n_slots, value):
        """later work on this context starts only when flags[0..n_slots) have all reached `value` (bounded device-side wait)"""
        N.check(self.handle, N.lib.lbvh_frame_wait(self.handle, flags_ptr, n_slots, value))


class DataBuffer:
    """Assets/_Scripts/DataBuffer.cs: a device buffer (ComputeBuffer) + a host mirror (T[])."""

    def __init__(self, ctx, size, dtype, initial_value=None):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.size = int(size)
        self.local = np.zeros(self.size, dtype=self.dtype)          # _localBuffer
        p = C.c_void_p()
        N.check(ctx.handle, N.lib.lbvh_buffer_alloc(ctx.handle, self.size, self.dtype.itemsize, C.byref(p)))
        self.device = p                                             # _deviceBuffer
        self._synced = False
        if initial_value is not None:                               # DataBuffer(size, initialValue) :14-23
            self.fill_u32(initial_value)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def fill_u32(self, word, mirror=True):
        """Every 32-bit word of the buffer = `word` (0xFFFFFFFF = NullLeaf / uint.MaxValue).
        mirror=False leaves the host copy alone (per-frame rebuilds)."""
        N.check(self.ctx.handle, N.lib.lbvh_buffer_fill_u32(self.ctx.handle, self.device, word, self.nbytes // 4))
        if mirror:
            self.local.view(np.uint32)[:] = word
        self._synced = mirror

    def get_data(self):                                             # GetData :50-54 (blocking)
        N.check(self.ctx.handle, N.lib.lbvh_buffer_download(
            self.ctx.handle, self.local.ctypes.data_as(C.c_void_p), self.device, self.nbytes))
        self._synced = True
        return self.local

    def sync(self):                                                 # Sync() = SetData :56-60
        N.check(self.ctx.handle, N.lib.lbvh_buffer_upload(
            self.ctx.handle, self.device, self.local.ctypes.data_as(C.c_void_p), self.nbytes))
        self._synced = True

    def dispose(self):                                              # Dispose :72-75
        if self.device:
            N.lib.lbvh_buffer_free(self.ctx.handle, self.device)
            self.device = None


class MeshBufferContainer:
    """Assets/_Scripts/MeshBufferContainer.cs.  The constructor takes the triangle soup (the
    reference's Mesh -> Triangle[] conversion, :117-146, stays on the caller's side) and runs the
    per-triangle Morton/AABB loop on the GPU instead of the CPU."""

    def __init__(self, ctx, triangles, capacity=None, box_min=L.SCENE_BOX_MIN, box_max=L.SCENE_BOX_MAX):
        triangles = np.ascontiguousarray(triangles, dtype=L.TRIANGLE)
        n = len(triangles)
        self.ctx = ctx
        self.triangles_length = n                                    # _trianglesLength
        self.capacity = capacity_for(n) if capacity is None else int(capacity)
        cap = self.capacity
        self.keys = DataBuffer(ctx, cap, np.uint32)                  # :108 (filled by morton_aabb)
        self.triangle_index = DataBuffer(ctx, cap, np.uint32)        # :109
        self.triangle_data = DataBuffer(ctx, cap, L.TRIANGLE)        # :110
        self.triangle_aabb = DataBuffer(ctx, cap, L.AABB)            # :111
        self.bvh_data = DataBuffer(ctx, cap, L.AABB)                 # :113
        self.bvh_leaf_node = DataBuffer(ctx, cap, L.LEAF_NODE, L.NULL)        # :114
        self.bvh_internal_node = DataBuffer(ctx, cap, L.INTERNAL_NODE, L.NULL)  # :115
        self.triangle_data.local[:n] = triangles
        self.triangle_data.sync()                                    # :150
        self.box_min = np.ascontiguousarray(box_min, dtype=np.float32)
        self.box_max = np.ascontiguousarray(box_max, dtype=np.float32)
        self.generate_keys()

    def generate_keys(self):
        """The loop of :123-146 + the Sync()s of :148-151 as one kernel."""
        f3 = C.POINTER(C.c_float)
        N.check(self.ctx.handle, N.lib.lbvh_morton_aabb(
            self.ctx.handle, self.triangle_data.device, self.triangles_length, self.capacity,
            self.box_min.ctypes.data_as(f3), self.box_max.ctypes.data_as(f3),
            self.keys.device, self.triangle_index.device, self.triangle_aabb.device))

    def distribute_keys(self):                                       # DistributeKeys :154-169
        N.check(self.ctx.handle, N.lib.lbvh_distribute_keys(self.ctx.handle, self.keys.device, self.triangles_length))

    def get_all_gpu_data(self):                                      # GetAllGpuData :171-196
        for b in (self.keys, self.triangle_index, self.triangle_data, self.triangle_aabb,
                  self.bvh_data, self.bvh_leaf_node, self.bvh_internal_node):
            b.get_data()
        n = self.triangles_length
        leaf, inner = self.bvh_leaf_node.local, self.bvh_internal_node.local
        bad_leaf = np.nonzero((leaf["index"][:n] == L.NULL) & (leaf["parent"][:n] == L.NULL))[0]
        bad_inner = np.nonzero((inner["index"][:n - 1] == L.NULL) & (inner["parent"][:n - 1] == L.NULL))[0]
        return bad_leaf, bad_inner                                   # "LEAF/INTERNAL CORRUPTED" :181-195

    def scene(self):
        s = N.Scene()
        s.n = self.triangles_length
        s.sorted_indices = self.triangle_index.device.value
        s.triangle_aabb = self.triangle_aabb.device.value
        s.internal_nodes = self.bvh_internal_node.device.value
        s.leaf_nodes = self.bvh_leaf_node.device.value
        s.bvh = self.bvh_data.device.value
        s.triangles = self.triangle_data.device.value
        return s

    def dispose(self):                                               # Dispose :207-216
        for b in (self.keys, self.triangle_index, self.triangle_data, self.triangle_aabb,
                  self.bvh_data, self.bvh_leaf_node, self.bvh_internal_node):
            b.dispose()


class ComputeBufferSorter:
    """Assets/_Scripts/ComputeBufferSorter.cs.  As in the reference, `data_length` (the triangle
    count, RaytracingMeshDrawer.cs:36) only bounds the sortedness validation (:150-177); Sort()
    always sorts the whole padded buffers (every dispatch covers DATA_ARRAY_COUNT, :107,116)."""

    def __init__(self, ctx, data_length, keys, values):
        self.ctx, self.data_length, self.keys, self.values = ctx, int(data_length), keys, values

    def sort(self):                                                  # Sort() :100-126
        N.check(self.ctx.handle, N.lib.lbvh_sort_pairs(
            self.ctx.handle, self.keys.device, self.values.device, self.keys.size))

    def validate_sorted_data(self):                                  # ValidateSortedData :150-177
        k = self.keys.get_data()[: self.data_length]
        return bool(np.all(k[1:] >= k[:-1])), int(np.count_nonzero(k[1:] == k[:-1]))

    def dispose(self):                                               # scratch lives in the context
        pass


class BVHConstructor:
    """Assets/_Scripts/BVHConstructor.cs."""

    def __init__(self, ctx, triangles_count, sorted_morton_codes, sorted_triangle_indices, triangle_aabb,
                 internal_nodes, leaf_nodes, bvh_data):
        self.ctx = ctx
        self.n = int(triangles_count)
        self.keys, self.indices, self.triangle_aabb = sorted_morton_codes, sorted_triangle_indices, triangle_aabb
        self.internal, self.leaf, self.bvh = internal_nodes, leaf_nodes, bvh_data

    def construct_tree(self):                                        # ConstructTree :61-64
        N.check(self.ctx.handle, N.lib.lbvh_build_tree(
            self.ctx.handle, self.n, self.keys.device, self.internal.device, self.leaf.device))

    def construct_bvh(self):                                         # ConstructBVH :66-69
        N.check(self.ctx.handle, N.lib.lbvh_refit(
            self.ctx.handle, self.n, self.internal.device, self.leaf.device, self.triangle_aabb.device,
            self.indices.device, self.bvh.device))

    def dispose(self):
        pass


class RaytracingMeshDrawer:
    """Assets/_Scripts/RaytracingMeshDrawer.cs: awake() = the build chain of Awake() :30-51,
    update() = the per-frame dispatch of Update() :76-84 (hit records instead of shaded pixels)."""

    def __init__(self, ctx, triangles, capacity=None):
        self.ctx = ctx
        self._triangles = triangles
        self._capacity = capacity
        self.container = None
        self._hits = None
        self._stats = None

    def awake(self, fast=True):
        ctx = self.ctx
        self.container = c = MeshBufferContainer(ctx, self._triangles, self._capacity)           # :34
        self.sorter = ComputeBufferSorter(ctx, c.triangles_length, c.keys, c.triangle_index)     # :36
        self.sorter.sort()                                                                       # :37
        c.distribute_keys()                                                                      # :39
        self.bvh_constructor = BVHConstructor(ctx, c.triangles_length, c.keys, c.triangle_index,
                                              c.triangle_aabb, c.bvh_internal_node, c.bvh_leaf_node,
                                              c.bvh_data)                                        # :41-48
        self.bvh_constructor.construct_tree()                                                    # :50
        self.bvh_constructor.construct_bvh()                                                     # :51
        if fast:
            self.build_fast_scene()
        return self

    def rebuild(self, fast=True, staged=False):
        """Per-frame rebuild on the same buffers (dynamic scenes): Morton -> ... -> refit (+ the derived traversal
        scene).  One lbvh_build_scene call (the two chains after the sort run concurrently); staged=True issues
        the reference's stage calls one by one instead — same results."""
        c = self.container
        if staged:
            c.bvh_leaf_node.fill_u32(L.NULL, mirror=False)
            c.bvh_internal_node.fill_u32(L.NULL, mirror=False)
            c.generate_keys()
            self.sorter.sort()
            c.distribute_keys()
            self.bvh_constructor.construct_tree()
            self.bvh_constructor.construct_bvh()
            if fast:
                self.build_fast_scene()
            return
        f3 = C.POINTER(C.c_float)
        N.check(self.ctx.handle, N.lib.lbvh_build_scene(
            self.ctx.handle, c.triangle_data.device, c.triangles_length, c.capacity, c.box_min.ctypes.data_as(f3),
            c.box_max.ctypes.data_as(f3), c.keys.device, c.triangle_index.device, c.triangle_aabb.device,
            c.bvh_internal_node.device, c.bvh_leaf_node.device, c.bvh_data.device,
            L.BUILD_RESET_NODES | (L.BUILD_FAST_SCENE if fast else 0)))

    def build_fast_scene(self):
        c = self.container
        s = c.scene()
        f3 = C.POINTER(C.c_float)
        N.check(self.ctx.handle, N.lib.lbvh_build_fast_scene(self.ctx.handle, C.byref(s), c.box_min.ctypes.data_as(f3),
                                                             c.box_max.ctypes.data_as(f3)))

    def update(self, camera, rect=None, mode=L.TRACE_FAST, stats=False):
        """Enqueue one frame (or the sub-rectangle (x0, y0, x1, y1) of it).  Returns the device
        hit buffer; read it back with hits()."""
        cam = N.Camera.from_dict(camera)
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, cam.screen_width, cam.screen_height)
        count = max((x1 - x0) * (y1 - y0), 1)
        if self._hits is None or self._hits.size < count:
            if self._hits is not None:
                self._hits.dispose()
            self._hits = DataBuffer(self.ctx, count, L.HIT)
        if stats and self._stats is None:
            self._stats = DataBuffer(self.ctx, 1, L.TRACE_STATS)
        self._rect = (x0, y0, x1, y1)
        s = self.container.scene()
        N.check(self.ctx.handle, N.lib.lbvh_trace_primary(
            self.ctx.handle, C.byref(cam), x0, y0, x1, y1, C.byref(s), mode, self._hits.device,
            self._stats.device if stats else None))
        return self._hits

    def update_shard(self, camera, shard_index, shard_count, mode=L.TRACE_FAST, stats=False):
        """One launch tracing this GPU's share of the full frame: every shard_count-th group of 8
        adjacent tiles.  The hit buffer is full-frame sized; only the shard's pixels are written."""
        cam = N.Camera.from_dict(camera)
        count = cam.screen_width * cam.screen_height
        if self._hits is None or self._hits.size < count:
            if self._hits is not None:
                self._hits.dispose()
            self._hits = DataBuffer(self.ctx, count, L.HIT)
        if stats and self._stats is None:
            self._stats = DataBuffer(self.ctx, 1, L.TRACE_STATS)
        self._rect = (0, 0, cam.screen_width, cam.screen_height)
        s = self.container.scene()
        N.check(self.ctx.handle, N.lib.lbvh_trace_primary_shard(
            self.ctx.handle, C.byref(cam), shard_index, shard_count, C.byref(s), mode, self._hits.device,
            self._stats.device if stats else None))
        return self._hits

    def hits(self):
        x0, y0, x1, y1 = self._rect
        return self._hits.get_data()[: (x1 - x0) * (y1 - y0)].reshape(y1 - y0, x1 - x0).copy()

    def set_texture(self, rgba8):
        """_objectDrawer.SetTexture("_meshTexture", ...) (Assets/_Scripts/RaytracingMeshDrawer.cs:61):
        (h, w, 4) uint8, row 0 at v = 0."""
        tex = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert tex.ndim == 3 and tex.shape[2] == 4
        self._tex_shape = tex.shape
        self._tex = DataBuffer(self.ctx, tex.size // 4, np.uint32)
        self._tex.local[:] = tex.reshape(-1, 4).view(np.uint32).reshape(-1)
        self._tex.sync()

    def shade(self):
        """The shading tail of the Raytracing kernel over the last update()'s hit records -> RGBA16F."""
        x0, y0, x1, y1 = self._rect
        count = (x1 - x0) * (y1 - y0)
        if getattr(self, "_image", None) is None or self._image.size < count:
            self._image = DataBuffer(self.ctx, count, np.uint64)       # 4 halves per pixel
        N.check(self.ctx.handle, N.lib.lbvh_shade(
            self.ctx.handle, self._hits.device, count, self.container.triangle_data.device, self._tex.device,
            self._tex_shape[1], self._tex_shape[0], self._image.device))
        return self._image

    def on_render_image(self, src):
        """OnRenderImage (Assets/_Scripts/RaytracingMeshDrawer.cs:86-90): Graphics.Blit(src, dest, _imageComposerMaterial) —
        the shaded image laid over the camera's own rendering `src` ((h, w, 4) float16); returns dest as float16."""
        x0, y0, x1, y1 = self._rect
        count = (x1 - x0) * (y1 - y0)
        bg = np.ascontiguousarray(src, dtype=np.float16)
        assert bg.shape == (y1 - y0, x1 - x0, 4)
        buf = DataBuffer(self.ctx, count, np.uint64)
        buf.local[:] = bg.reshape(-1, 4).view(np.uint64).reshape(-1)
        buf.sync()
        N.check(self.ctx.handle, N.lib.lbvh_compose(self.ctx.handle, buf.device, self._image.device, count, buf.device))
        out = buf.get_data()[:count].view(np.float16).reshape(y1 - y0, x1 - x0, 4).copy()
        buf.dispose()
        return out

    def image(self):
        x0, y0, x1, y1 = self._rect
        n = (x1 - x0) * (y1 - y0)
        return self._image.get_data()[:n].view(np.float16).reshape(y1 - y0, x1 - x0, 4).copy()

    def stats(self):
        return self._stats.get_data()[0].copy()

    def on_destroy(self):                                            # OnDestroy :118-123
        if self.container:
            self.container.dispose()
        for b in (self._hits, self._stats):
            if b is not None:
                b.dispose()


class MultiGpuDrawer:
    """One frame from N GPUs driven by one process (twin of host/lbvh_host.hpp MultiGpuDrawer; BASELINE configs[2]): one
    Context + one replica of the scene per entry of `devices` (a device may repeat: logical ranks on one GPU), every build
    call enqueued round-robin, and update() = every context traces its share straight into ONE full-frame buffer on the
    first device (peer-mapped stores, no second pass) + the first context's stream waits for the others' completion events
    on the device.  The reference renders one image per Update() (Assets/_Scripts/RaytracingMeshDrawer.cs:76-89): this is
    where the N shares become that image."""

    def __init__(self, devices, triangles, capacity=None):
        self.contexts = [Context(d) for d in devices]
        self.drawers = [RaytracingMeshDrawer(c, triangles, capacity) for c in self.contexts]
        self.done = [c.sync_event() for c in self.contexts]
        self.consumed = self.contexts[0].sync_event()
        for c in self.contexts[1:]:
            c.peer_enable(devices[0])
        self.frame = None
        self._shape = None

    @property
    def owner(self):
        return self.contexts[0]

    def awake(self, fast=True):
        for d in self.drawers:                       # replicas: the same deterministic build on every GPU
            d.awake(fast=fast)
        return self

    def rebuild(self, fast=True):
        for d in self.drawers:
            d.rebuild(fast=fast)

    def update(self, camera, mode=L.TRACE_FAST):
        cam = N.Camera.from_dict(camera)
        count = cam.screen_width * cam.screen_height
        if self.frame is None or self.frame.size < count:
            self.sync()                              # nobody may still be writing the old buffer
            if self.frame is not None:
                self.frame.dispose()
            self.frame = DataBuffer(self.owner, count, L.HIT)
        self._shape = (cam.screen_height, cam.screen_width)
        n = len(self.contexts)
        self.owner.record(self.consumed)             # the owner's reads of the previous frame end here
        for c in self.contexts[1:]:
            c.wait_event(self.consumed)
        for r, (c, d) in enumerate(zip(self.contexts, self.drawers)):
            s = d.container.scene()
            N.check(c.handle, N.lib.lbvh_trace_primary_shard(c.handle, C.byref(cam), r, n, C.byref(s), mode, self.frame.device, None))
            if r:
                c.record(self.done[r])
        for r in range(1, n):
            self.owner.wait_event(self.done[r])      # the gather: a device-side wait, nothing is copied
        return self.frame

    def hits(self):
        h, w = self._shape
        return self.frame.get_data()[: h * w].reshape(h, w).copy()

    def sync(self):
        for c in self.contexts:
            c.sync()

    def on_destroy(self):
        self.sync()
        if self.frame is not None:
            self.frame.dispose()
        for d in self.drawers:
            d.on_destroy()
        for c, e in zip(self.contexts, self.done):
            c.destroy_event(e)
        self.owner.destroy_event(self.consumed)
        for c in self.contexts:
            c.close()


class DynamicPathTracer:
    """SURVEY 8(f) rank 3 / BASELINE configs[4] (extension, no reference counterpart): per frame the rigid bodies
    are rotated (lbvh_animate), the whole LBVH is rebuilt on the same buffers (RaytracingMeshDrawer.rebuild), primary
    rays go through the packet kernel and `bounces` diffuse bounces through lbvh_trace_rays / lbvh_path_scatter."""

    def __init__(self, ctx, rest_triangles, body_ids, body_centres, t_min=1e-3, albedo=0.7, seed=1):
        self.ctx = ctx
        self.drawer = RaytracingMeshDrawer(ctx, rest_triangles).awake()
        n = self.drawer.container.triangles_length
        self.rest = DataBuffer(ctx, n, L.TRIANGLE)
        self.rest.local[:] = np.ascontiguousarray(rest_triangles, dtype=L.TRIANGLE)
        self.rest.sync()
        self.body = DataBuffer(ctx, n, np.uint32)
        self.body.local[:] = np.ascontiguousarray(body_ids, dtype=np.uint32)
        self.body.sync()
        ctr = np.ascontiguousarray(body_centres, dtype=np.float32).reshape(-1, 4)
        self.centres = DataBuffer(ctx, ctr.size, np.float32)
        self.centres.local[:] = ctr.reshape(-1)
        self.centres.sync()
        self.t_min, self.albedo, self.seed = float(t_min), float(albedo), int(seed)
        self.states = self.hits = self.image_buf = None

    def animate(self, angle, fused=True):
        """the bodies turned to `angle` and the whole LBVH rebuilt: lbvh_animate_build_scene (one call; fused=False: lbvh_animate +
        the rebuild as two — same results)"""
        c = self.drawer.container
        cs, sn = float(np.float32(np.cos(angle))), float(np.float32(np.sin(angle)))
        if not fused:
            N.check(self.ctx.handle, N.lib.lbvh_animate(
                self.ctx.handle, self.rest.device, c.triangles_length, self.body.device, self.centres.device, cs, sn, c.triangle_data.device))
            self.drawer.rebuild(fast=True)
            return
        f3 = C.POINTER(C.c_float)
        N.check(self.ctx.handle, N.lib.lbvh_animate_build_scene(
            self.ctx.handle, self.rest.device, self.body.device, self.centres.device, cs, sn, c.triangle_data.device, c.triangles_length,
            c.capacity, c.box_min.ctypes.data_as(f3), c.box_max.ctypes.data_as(f3), c.keys.device, c.triangle_index.device,
            c.triangle_aabb.device, c.bvh_internal_node.device, c.bvh_leaf_node.device, c.bvh_data.device,
            L.BUILD_RESET_NODES | L.BUILD_FAST_SCENE))

    def render(self, camera, bounces=4):
        cam = N.Camera.from_dict(camera)
        count = cam.screen_width * cam.screen_height
        if self.states is None or self.states.size < count:
            self.states = DataBuffer(self.ctx, count, L.PATH_STATE)
            self.hits = DataBuffer(self.ctx, count, L.HIT)
            self.image_buf = DataBuffer(self.ctx, count, np.uint64)
        h = self.ctx.handle
        s = self.drawer.container.scene()
        # primary rays: the coherent packet kernel — BEFORE the path states are initialised: 132 MB of state stores right
        # in front of it push the scene out of the caches (the frame's primary trace 0.274 -> 0.253 ms)
        N.check(h, N.lib.lbvh_trace_primary(h, C.byref(cam), 0, 0, cam.screen_width, cam.screen_height, C.byref(s),
                                            L.TRACE_FAST, self.hits.device, None))
        # bounce b = scatter at the hits of segment b + trace of segment b + 1 (one call); the first one also makes the path
        # states from the camera (lbvh_path_first_bounce = lbvh_path_begin + bounce 0); the last bounce only scatters
        if bounces == 0:
            N.check(h, N.lib.lbvh_path_begin(h, C.byref(cam), self.states.device))
        else:
            N.check(h, N.lib.lbvh_path_first_bounce(h, C.byref(cam), C.byref(s), self.states.device, self.hits.device, self.seed,
                                                    self.albedo, self.t_min))
        for b in range(1, bounces):
            N.check(h, N.lib.lbvh_path_bounce(h, C.byref(s), self.states.device, self.hits.device, count, b, self.seed,
                                              self.albedo, self.t_min))
        N.check(h, N.lib.lbvh_path_scatter(h, C.byref(s), self.hits.device, count, bounces, self.seed, self.albedo,
                                           self.states.device))
        N.check(h, N.lib.lbvh_path_resolve(h, self.states.device, count, self.image_buf.device))
        self._shape = (cam.screen_height, cam.screen_width)
        return self.image_buf

    def image(self):
        n = self._shape[0] * self._shape[1]
        return self.image_buf.get_data()[:n].view(np.float16).reshape(self._shape + (4,)).copy()
