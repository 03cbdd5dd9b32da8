// BVHConstructor.Native.cs — Assets/_Scripts/BVHConstructor.cs re-hosted on liblbvh.so.
//
// Same constructor (trianglesCount, sortedMortonCodes, sortedTriangleIndices, triangleAABB, internalNodes, leafNodes,
// BVHData, container) and ConstructTree() / ConstructBVH() / Dispose() as the reference:
//   ConstructTree -> lbvh_build_tree  (kernel TreeConstructor, BVH.compute:94-149: Karras topology, bit-exact node words)
//   ConstructBVH  -> lbvh_refit       (kernel BVHConstructor,  BVH.compute:172-220: bottom-up AABB union)
// The per-node arrival flags (`_atomics`, BVHConstructor.cs:41) belong to the native context, which clears them per call.
// `container` is accepted and ignored; it may be null.
// SOURCE ONLY (no C# toolchain in the build image); surface checked by tests/test_csharp_surface.py.
using System;

public class BVHConstructor : IDisposable
{
    readonly NativeBuffer _sortedMortonCodes;
    readonly NativeBuffer _sortedTriangleIndices;
    readonly NativeBuffer _triangleAABB;
    readonly NativeBuffer _internalNodes;
    readonly NativeBuffer _leafNodes;
    readonly NativeBuffer _bvhData;
    readonly uint _trianglesCount;

    public BVHConstructor(
        uint trianglesCount,
        NativeBuffer sortedMortonCodes,
        NativeBuffer sortedTriangleIndices,
        NativeBuffer triangleAABB,
        NativeBuffer internalNodes,
        NativeBuffer leafNodes,
        NativeBuffer BVHData,
        IShaderContainer container)
    {
        if (trianglesCount < 2) throw new ArgumentException("BVHConstructor: at least 2 triangles (the reference underflows n - 1)");
        _trianglesCount = trianglesCount;
        _sortedMortonCodes = sortedMortonCodes;
        _sortedTriangleIndices = sortedTriangleIndices;
        _triangleAABB = triangleAABB;
        _internalNodes = internalNodes;
        _leafNodes = leafNodes;
        _bvhData = BVHData;
    }

    public void ConstructTree()
    {
        IntPtr ctx = _internalNodes.Context;             // the GPU the tree's buffers live on
        LbvhNative.Check(ctx, LbvhNative.lbvh_build_tree(ctx, _trianglesCount, _sortedMortonCodes.Pointer, _internalNodes.Pointer,
                                                         _leafNodes.Pointer));
    }

    public void ConstructBVH()
    {
        IntPtr ctx = _internalNodes.Context;
        LbvhNative.Check(ctx, LbvhNative.lbvh_refit(ctx, _trianglesCount, _internalNodes.Pointer, _leafNodes.Pointer, _triangleAABB.Pointer,
                                                    _sortedTriangleIndices.Pointer, _bvhData.Pointer));
    }

    public void Dispose()
    {
        // the arrival flags live in the native context (freed by LbvhContext.Shutdown)
    }
}
