// LbvhNative.cs — P/Invoke binding of liblbvh.so (include/lbvh.h) for the reference's C# host.
//
// SOURCE ONLY: this image has no C# toolchain (no dotnet/mono/csc), so this file is not compiled here;
// tests/test_abi.py checks every declaration's name and arity against include/lbvh.h.  It is the shim a
// maintainer of drzhn/UnitySimpleRaytracing adds under Assets/_Scripts/ together with the re-hosted classes
// next to it (DataBuffer / MeshBufferContainer / ComputeBufferSorter / BVHConstructor / RaytracingMeshDrawer
// `.Native.cs`, which replace the reference files of the same class names; see INTEGRATION.md).
// Struct layouts are the reference's own Sequential/Pack=16 structs (SceneDataTypes.cs), which are
// already byte-identical to lbvh_triangle / lbvh_aabb / lbvh_internal_node / lbvh_leaf_node.
using System;
using System.Runtime.InteropServices;

public static class LbvhNative
{
    const string Lib = "lbvh";   // liblbvh.so on Linux

    public const int ABI_VERSION = 11;             // LBVH_ABI_VERSION of the include/lbvh.h this file was written against
    public const int TRACE_REFERENCE = 0, TRACE_FAST = 1;
    // TRACE_FAST + the reference's choice wherever two triangles are hit at exactly the same t: every record == TRACE_REFERENCE's
    // (both fast modes: a computed t in front of its own triangle's box — fp32 noise on a grazing ray, which the un-pruned reference
    // loop reports — does not count: include/lbvh.h at the traversal flavours, DESIGN 2.4)
    public const int TRACE_FAST_EXACT = 2;
    public const uint BUILD_FAST_SCENE = 1, BUILD_RESET_NODES = 2;      // lbvh_build_scene flags

    [StructLayout(LayoutKind.Sequential)]
    public struct Hit { public float t; public uint tri; public float u, v; }

    // lbvh_camera (include/lbvh.h): the 16 matrix floats are plain fields, row-major m00..m33 as Unity's Matrix4x4 names
    // them, so the struct needs no `unsafe` / "allow unsafe code" project setting and marshals by value as it is.
    [StructLayout(LayoutKind.Sequential)]
    public struct Camera
    {
        public int screenWidth, screenHeight;
        public float cameraFov, nearPlane;
        public float m00, m01, m02, m03, m10, m11, m12, m13, m20, m21, m22, m23, m30, m31, m32, m33;
    }

    [StructLayout(LayoutKind.Sequential, CharSet = CharSet.Ansi)]
    public struct ProfileRow
    {
        [MarshalAs(UnmanagedType.ByValTStr, SizeConst = 48)] public string name;     // char name[48]
        public uint launches;
        public float totalMs;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct Scene
    {
        public uint n;
        public IntPtr sortedIndices, triangleAabb, internalNodes, leafNodes, bvh, triangles;
    }

    [DllImport(Lib)] public static extern int lbvh_abi_version();
    [DllImport(Lib)] public static extern int lbvh_device_count();
    [DllImport(Lib)] public static extern int lbvh_create(int deviceId, out IntPtr ctx);
    [DllImport(Lib)] public static extern int lbvh_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr lbvh_last_error(IntPtr ctx);
    [DllImport(Lib)] public static extern int lbvh_sync(IntPtr ctx);

    [DllImport(Lib)] public static extern int lbvh_buffer_alloc(IntPtr ctx, UIntPtr count, UIntPtr stride, out IntPtr dPtr);
    [DllImport(Lib)] public static extern int lbvh_buffer_free(IntPtr ctx, IntPtr dPtr);
    [DllImport(Lib)] public static extern int lbvh_buffer_fill_u32(IntPtr ctx, IntPtr dPtr, uint value, UIntPtr nWords);
    [DllImport(Lib)] public static extern int lbvh_buffer_upload(IntPtr ctx, IntPtr dDst, IntPtr hSrc, UIntPtr bytes);
    [DllImport(Lib)] public static extern int lbvh_buffer_download(IntPtr ctx, IntPtr hDst, IntPtr dSrc, UIntPtr bytes);

    [DllImport(Lib)] public static extern int lbvh_morton_aabb(IntPtr ctx, IntPtr dTriangles, uint n, uint capacity,
        float[] boxMin, float[] boxMax, IntPtr dKeys, IntPtr dIndices, IntPtr dAabb);
    [DllImport(Lib)] public static extern int lbvh_sort_pairs(IntPtr ctx, IntPtr dKeys, IntPtr dValues, uint count);
    [DllImport(Lib)] public static extern int lbvh_distribute_keys(IntPtr ctx, IntPtr dKeys, uint n);
    [DllImport(Lib)] public static extern int lbvh_build_tree(IntPtr ctx, uint n, IntPtr dSortedKeys, IntPtr dInternal, IntPtr dLeaf);
    [DllImport(Lib)] public static extern int lbvh_refit(IntPtr ctx, uint n, IntPtr dInternal, IntPtr dLeaf,
        IntPtr dTriangleAabb, IntPtr dSortedIndices, IntPtr dBvh);
    [DllImport(Lib)] public static extern int lbvh_build_fast_scene(IntPtr ctx, ref Scene scene, float[] boxMin, float[] boxMax);
    [DllImport(Lib)] public static extern int lbvh_trace_primary(IntPtr ctx, ref Camera camera, int x0, int y0, int x1, int y1,
        ref Scene scene, int mode, IntPtr dHits, IntPtr dStats);

    [DllImport(Lib)] public static extern int lbvh_trace_primary_shard(IntPtr ctx, ref Camera camera, uint shardIndex, uint shardCount,
        ref Scene scene, int mode, IntPtr dHits, IntPtr dStats);

    [DllImport(Lib)] public static extern int lbvh_shade(IntPtr ctx, IntPtr dHits, UIntPtr count, IntPtr dTriangles,
        IntPtr dTextureRgba8, int texW, int texH, IntPtr dRgba16f);
    [DllImport(Lib)] public static extern int lbvh_compose(IntPtr ctx, IntPtr dBackgroundRgba16f, IntPtr dObjectRgba16f, UIntPtr count,
                                                            IntPtr dOutRgba16f);

    // the whole Awake() build chain in one call (per-frame rebuilds); flags: 1 = fast scene, 2 = reset node arrays
    [DllImport(Lib)] public static extern int lbvh_build_scene(IntPtr ctx, IntPtr dTriangles, uint n, uint capacity, float[] boxMin,
        float[] boxMax, IntPtr dKeys, IntPtr dIndices, IntPtr dAabb, IntPtr dInternal, IntPtr dLeaf, IntPtr dBvh, uint flags);

    // LBVH_TRACE_FAST keeps a dispatch hint from the previous frame; this drops it (the next frame runs as a first frame)
    [DllImport(Lib)] public static extern int lbvh_trace_forget(IntPtr ctx);
    // multi-GPU frames: this context's per-tile costs into a full-frame array / the merged array of all ranks back
    [DllImport(Lib)] public static extern int lbvh_trace_costs_export(IntPtr ctx, IntPtr dFrameCosts, uint tilesX, uint tilesY);
    [DllImport(Lib)] public static extern int lbvh_trace_costs_import(IntPtr ctx, IntPtr dFrameCosts, uint tilesX, uint tilesY);

    // one frame from N GPUs (BASELINE configs[2]): peer-mapped frame buffer, ordering between contexts of this process
    // (sync events) or between processes (IPC handle of the buffer + completion flags waited for on the device), and
    // packed shares for transports that want one contiguous block per GPU
    [DllImport(Lib)] public static extern int lbvh_peer_enable(IntPtr ctx, int peerDevice);
    [DllImport(Lib)] public static extern int lbvh_sync_event_create(IntPtr ctx, out IntPtr ev);
    [DllImport(Lib)] public static extern int lbvh_event_wait(IntPtr ctx, IntPtr ev);
    [DllImport(Lib)] public static extern int lbvh_ipc_export(IntPtr ctx, IntPtr dPtr, [Out] byte[] handle64);
    [DllImport(Lib)] public static extern int lbvh_ipc_import(IntPtr ctx, byte[] handle64, out IntPtr dPtr);
    [DllImport(Lib)] public static extern int lbvh_ipc_close(IntPtr ctx, IntPtr dPtr);
    // completion flags a running kernel may poll while another GPU / process stores into them: uncached device memory
    [DllImport(Lib)] public static extern int lbvh_flags_alloc(IntPtr ctx, UIntPtr nWords, out IntPtr dFlags);
    [DllImport(Lib)] public static extern int lbvh_frame_signal(IntPtr ctx, IntPtr dFlags, uint slot, uint value);
    [DllImport(Lib)] public static extern int lbvh_frame_wait(IntPtr ctx, IntPtr dFlags, uint nSlots, uint value);
    [DllImport(Lib)] public static extern int lbvh_trace_primary_shard_packed(IntPtr ctx, ref Camera camera, uint shardIndex, uint shardCount,
        ref Scene scene, int mode, IntPtr dPacked, IntPtr dStats);
    [DllImport(Lib)] public static extern ulong lbvh_shard_records(int width, int height, uint shardIndex, uint shardCount);
    [DllImport(Lib)] public static extern int lbvh_frame_unpack(IntPtr ctx, IntPtr dPacked, ulong shareStride, uint firstShard, uint nShards,
        uint shardCount, int width, int height, IntPtr dFrameHits);
    // a context whose work is ordered by a stream the caller owns (hipStream_t), e.g. an interop stream
    [DllImport(Lib)] public static extern int lbvh_create_on_stream(int deviceId, IntPtr hipStream, out IntPtr ctx);
    // HIP events on the context's stream
    [DllImport(Lib)] public static extern int lbvh_event_create(IntPtr ctx, out IntPtr ev);
    [DllImport(Lib)] public static extern int lbvh_event_destroy(IntPtr ctx, IntPtr ev);
    [DllImport(Lib)] public static extern int lbvh_event_record(IntPtr ctx, IntPtr ev);
    [DllImport(Lib)] public static extern int lbvh_event_elapsed_ms(IntPtr ctx, IntPtr start, IntPtr stop, out float ms);

    // dynamic scene + secondary rays (BASELINE configs[4]; extension, no reference counterpart)
    [DllImport(Lib)] public static extern int lbvh_animate(IntPtr ctx, IntPtr dRestTriangles, uint n, IntPtr dBodyIds, IntPtr dBodyCentres,
        float cosAngle, float sinAngle, IntPtr dTrianglesOut);
    [DllImport(Lib)] public static extern int lbvh_animate_build_scene(IntPtr ctx, IntPtr dRestTriangles, IntPtr dBodyIds, IntPtr dBodyCentres,
        float cosAngle, float sinAngle, IntPtr dTriangles, uint n, uint capacity, float[] boxMin, float[] boxMax, IntPtr dKeys, IntPtr dIndices,
        IntPtr dAabb, IntPtr dInternal, IntPtr dLeaf, IntPtr dBvh, uint flags);
    [DllImport(Lib)] public static extern int lbvh_path_begin(IntPtr ctx, ref Camera camera, IntPtr dStates);
    [DllImport(Lib)] public static extern int lbvh_trace_rays(IntPtr ctx, IntPtr dStates, UIntPtr count, float tMin, ref Scene scene, IntPtr dHits);
    [DllImport(Lib)] public static extern int lbvh_path_scatter(IntPtr ctx, ref Scene scene, IntPtr dHits, UIntPtr count, uint bounce, uint seed,
        float albedo, IntPtr dStates);
    [DllImport(Lib)] public static extern int lbvh_path_bounce(IntPtr ctx, ref Scene scene, IntPtr dStates, IntPtr dHits, UIntPtr count,
        uint bounce, uint seed, float albedo, float tMin);
    [DllImport(Lib)] public static extern int lbvh_path_first_bounce(IntPtr ctx, ref Camera camera, ref Scene scene, IntPtr dStates, IntPtr dHits,
        uint seed, float albedo, float tMin);
    [DllImport(Lib)] public static extern int lbvh_path_resolve(IntPtr ctx, IntPtr dStates, UIntPtr count, IntPtr dRgba16f);

    // local kernels of the multi-GPU key-range sharded sort (BASELINE configs[3])
    [DllImport(Lib)] public static extern int lbvh_key_histogram(IntPtr ctx, IntPtr dKeys, uint count, uint[] prefixes, uint nPrefixes,
        uint prefixShift, uint shift, IntPtr dHist);
    [DllImport(Lib)] public static extern int lbvh_lower_bound(IntPtr ctx, IntPtr dSortedKeys, uint count, uint[] probes, uint nProbes,
        IntPtr dPositions);

    // the same two with prefixes / probes in device memory (the sharded sort's splitter search stays on the GPU)
    [DllImport(Lib)] public static extern int lbvh_key_histogram_device(IntPtr ctx, IntPtr dKeys, uint count, IntPtr dPrefixes, uint nPrefixes,
        uint prefixShift, uint shift, IntPtr dHist);
    [DllImport(Lib)] public static extern int lbvh_lower_bound_device(IntPtr ctx, IntPtr dSortedKeys, uint count, IntPtr dProbes, uint nProbes,
        IntPtr dPositions);

    public static void Check(IntPtr ctx, int status)
    {
        if (status != 0)
            throw new InvalidOperationException($"lbvh status {status}: {Marshal.PtrToStringAnsi(lbvh_last_error(ctx))}");
    }
}
