// MeshBufferContainer.Native.cs — Assets/_Scripts/MeshBufferContainer.cs re-hosted on liblbvh.so.
//
// Same class name, constructor and public members as the reference (NativeBuffer where it says ComputeBuffer); it takes
// the reference file's place in a Unity project.  What changes is where the work happens:
//   * the constructor still gathers Triangle[] from the Mesh on the CPU (MeshBufferContainer.cs:117-146 — Unity's Mesh
//     lives in managed memory), uploads the 128-byte records ONCE, and then Morton codes, identity indices, padded AABBs
//     and the 0xFFFFFFFF pad slots come from one device pass (lbvh_morton_aabb) instead of the per-triangle C# loop
//     (:123-146, helpers :32-83) and its four full-capacity uploads (:148-151);
//   * DistributeKeys() is a device scan (lbvh_distribute_keys) instead of read-back -> serial loop -> upload (:154-169).
// The local mirrors (KeysData, TriangleAABBLocalData ...) are filled by GetAllGpuData() as in the reference.
// SOURCE ONLY (no C# toolchain in the build image); surface checked by tests/test_csharp_surface.py.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

public class MeshBufferContainer : IDisposable
{
    // the reference's scene box: Whole = [-125, 125]^3 (MeshBufferContainer.cs:9-15)
    static readonly float[] WholeMin = { -125f, -125f, -125f };
    static readonly float[] WholeMax = { 125f, 125f, 125f };

    public NativeBuffer Keys => _keysBuffer.DeviceBuffer;
    public uint[] KeysData => _keysBuffer.LocalBuffer;
    public NativeBuffer TriangleIndex => _triangleIndexBuffer.DeviceBuffer;
    public NativeBuffer TriangleData => _triangleDataBuffer.DeviceBuffer;
    public NativeBuffer TriangleAABB => _triangleAABBBuffer.DeviceBuffer;
    public NativeBuffer BvhData => _bvhDataBuffer.DeviceBuffer;
    public NativeBuffer BvhLeafNode => _bvhLeafNodesBuffer.DeviceBuffer;
    public NativeBuffer BvhInternalNode => _bvhInternalNodesBuffer.DeviceBuffer;

    public AABB[] TriangleAABBLocalData => _triangleAABBBuffer.LocalBuffer;
    public AABB[] BVHLocalData => _bvhDataBuffer.LocalBuffer;
    public LeafNode[] BvhLeafNodeLocalData => _bvhLeafNodesBuffer.LocalBuffer;
    public InternalNode[] BvhInternalNodeLocalData => _bvhInternalNodesBuffer.LocalBuffer;
    public uint TrianglesLength => _trianglesLength;

    readonly uint _trianglesLength;
    readonly DataBuffer<uint> _keysBuffer;
    readonly DataBuffer<uint> _triangleIndexBuffer;
    readonly DataBuffer<Triangle> _triangleDataBuffer;
    readonly DataBuffer<AABB> _triangleAABBBuffer;
    readonly DataBuffer<AABB> _bvhDataBuffer;
    readonly DataBuffer<LeafNode> _bvhLeafNodesBuffer;
    readonly DataBuffer<InternalNode> _bvhInternalNodesBuffer;

    public MeshBufferContainer(Mesh mesh)
    {
        // the ABI is these two sizes (MeshBufferContainer.cs:98-106 logs an error; here the native side would misread)
        if (Marshal.SizeOf(typeof(Triangle)) != 128 || Marshal.SizeOf(typeof(AABB)) != 32)
            throw new InvalidOperationException("Triangle must marshal to 128 bytes and AABB to 32");

        int capacity = Constants.DATA_ARRAY_COUNT;
        _keysBuffer = new DataBuffer<uint>(capacity);                    // pads written by lbvh_morton_aabb below
        _triangleIndexBuffer = new DataBuffer<uint>(capacity);
        _triangleDataBuffer = new DataBuffer<Triangle>(capacity);
        _triangleAABBBuffer = new DataBuffer<AABB>(capacity);
        _bvhDataBuffer = new DataBuffer<AABB>(capacity);
        _bvhLeafNodesBuffer = new DataBuffer<LeafNode>(capacity);
        _bvhInternalNodesBuffer = new DataBuffer<InternalNode>(capacity);
        _bvhLeafNodesBuffer.DeviceBuffer.Fill(0xFFFFFFFFu);              // LeafNode.NullLeaf / InternalNode.NullLeaf in
        _bvhInternalNodesBuffer.DeviceBuffer.Fill(0xFFFFFFFFu);          // every slot (SceneDataTypes.cs:63-71, 85-89)
        _bvhDataBuffer.DeviceBuffer.Fill(0u);
        _triangleAABBBuffer.DeviceBuffer.Fill(0u);

        Vector3[] vertices = mesh.vertices;
        int[] corners = mesh.triangles;
        Vector2[] uvs = mesh.uv;
        Vector3[] normals = mesh.normals;
        _trianglesLength = (uint)corners.Length / 3;
        if (_trianglesLength < 2 || _trianglesLength > (uint)capacity)
            throw new ArgumentException($"mesh has {_trianglesLength} triangles; the builder takes 2 .. {capacity}");

        Triangle[] tris = _triangleDataBuffer.LocalBuffer;
        for (uint i = 0; i < _trianglesLength; i++)
        {
            int ia = corners[i * 3 + 0], ib = corners[i * 3 + 1], ic = corners[i * 3 + 2];
            tris[i] = new Triangle
            {
                a = vertices[ia], b = vertices[ib], c = vertices[ic],
                a_uv = uvs[ia], b_uv = uvs[ib], c_uv = uvs[ic],
                a_normal = normals[ia], b_normal = normals[ib], c_normal = normals[ic],
            };
        }
        _triangleDataBuffer.Sync();

        IntPtr ctx = Keys.Context;                       // the GPU that was current when the buffers above were made
        LbvhNative.Check(ctx, LbvhNative.lbvh_morton_aabb(ctx, TriangleData.Pointer, _trianglesLength, (uint)capacity, WholeMin, WholeMax,
                                                          Keys.Pointer, TriangleIndex.Pointer, TriangleAABB.Pointer));
    }

    public void DistributeKeys()
    {
        IntPtr ctx = Keys.Context;
        LbvhNative.Check(ctx, LbvhNative.lbvh_distribute_keys(ctx, Keys.Pointer, _trianglesLength));
    }

    public void GetAllGpuData()
    {
        _keysBuffer.GetData();
        _triangleIndexBuffer.GetData();
        _triangleDataBuffer.GetData();
        _triangleAABBBuffer.GetData();
        _bvhDataBuffer.GetData();
        _bvhLeafNodesBuffer.GetData();
        _bvhInternalNodesBuffer.GetData();

        // the reference's coverage check (:181-195): every leaf / internal node below n was written by the tree kernel
        LeafNode[] leaves = _bvhLeafNodesBuffer.LocalBuffer;
        for (uint i = 0; i < _trianglesLength; i++)
            if (leaves[i].index == 0xFFFFFFFF && leaves[i].parent == 0xFFFFFFFF)
                Debug.LogErrorFormat("LEAF CORRUPTED {0}", i);
        InternalNode[] inner = _bvhInternalNodesBuffer.LocalBuffer;
        for (uint i = 0; i + 1 < _trianglesLength; i++)
            if (inner[i].index == 0xFFFFFFFF && inner[i].parent == 0xFFFFFFFF)
                Debug.LogErrorFormat("INTERNAL CORRUPTED {0}", i);
    }

    public void PrintData()
    {
        Debug.Log(_keysBuffer);
        Debug.Log(_bvhInternalNodesBuffer);
        Debug.Log(_bvhLeafNodesBuffer);
        Debug.Log(_bvhDataBuffer);
    }

    /// The scene descriptor the trace / shade calls take (the six buffers RaytracingMeshDrawer.cs:65-70 binds).
    public LbvhNative.Scene NativeScene() => new LbvhNative.Scene
    {
        n = _trianglesLength,
        sortedIndices = TriangleIndex.Pointer, triangleAabb = TriangleAABB.Pointer, internalNodes = BvhInternalNode.Pointer,
        leafNodes = BvhLeafNode.Pointer, bvh = BvhData.Pointer, triangles = TriangleData.Pointer,
    };

    public void Dispose()
    {
        _keysBuffer.Dispose();
        _triangleIndexBuffer.Dispose();
        _triangleDataBuffer.Dispose();
        _triangleAABBBuffer.Dispose();
        _bvhDataBuffer.Dispose();
        _bvhLeafNodesBuffer.Dispose();
        _bvhInternalNodesBuffer.Dispose();
    }
}
