// RaytracingMeshDrawer.Native.cs — Assets/_Scripts/RaytracingMeshDrawer.cs re-hosted on liblbvh.so.
//
// Same MonoBehaviour, serialized fields (so Scene.unity keeps its wiring) and Awake / Update / OnRenderImage / OnDestroy.
// Awake runs the reference's build chain through the re-hosted classes (RaytracingMeshDrawer.cs:34-51) and then builds the
// derived traversal scene LBVH_TRACE_FAST walks.  Update replaces the four uniforms + Dispatch (:78-83) by
// lbvh_trace_primary (hit records) + lbvh_shade (the kernel's last lines, Raytracing.compute:178-184 -> RGBA16F).
//
// Display: the reference's kernel writes a RenderTexture that ImageComposer.shader blends over the camera image
// (:56-73, :86-90).  HIP memory is not a D3D/Vulkan resource Unity can bind, so the RGBA16F frame comes back through
// host memory (8 bytes x W x H = 16.6 MB at 1080p; the hit records themselves never leave the GPU) into a Texture2D that
// takes the RenderTexture's place as `_ObjectTexture`.  Zero-copy display would need external-memory interop between the
// Vulkan device Unity renders with and HIP (hipImportExternalMemory) — not built.
//
// Several GPUs (BASELINE configs[2]; `_gpuDevices` in the inspector, empty = GPU 0 alone): Awake runs the build chain once per
// device — a replica of the scene on each, the same deterministic build — and Update has every GPU trace its share of the
// frame (lbvh_trace_primary_shard) STRAIGHT INTO the first GPU's hit buffer (peer-mapped stores: lbvh_peer_enable), the first
// GPU's stream waits for the others' completion events on the device (lbvh_sync_event_create / lbvh_event_wait), shades and
// hands the one image to Unity — the same one-frame-per-Update the reference renders (RaytracingMeshDrawer.cs:76-89), no
// collective, no copy.  Twin of host/lbvh_host.hpp MultiGpuDrawer (tested in tests/test_multi_gpu.py).
// SOURCE ONLY (no C# toolchain in the build image); surface checked by tests/test_csharp_surface.py.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

[RequireComponent(typeof(Camera))]
public class RaytracingMeshDrawer : MonoBehaviour
{
    [SerializeField] [Range(0, 6365)] private int _indexToCheck;
    [SerializeField] private Shader _imageComposer;
    [SerializeField] private Mesh _mesh;
    [SerializeField] private ShaderContainer _shaderContainer;      // unused by the native path; kept for the scene's wiring
    [SerializeField] private Texture _meshTexture;                  // must be a readable Texture2D
    [SerializeField] private int[] _gpuDevices;                     // HIP devices to shard the rays over; empty: device 0
    // true: LBVH_TRACE_FAST_EXACT — every hit record is the reference kernel's, also where two triangles are hit at exactly the
    // same t (a handful of pixels per frame; + 20 % traversal time).  false: LBVH_TRACE_FAST (same t; lowest triangle index there)
    [SerializeField] private bool _exactTies = true;

    private Camera _camera;
    private MeshBufferContainer _container;                         // rank 0's (the reference's fields)
    private ComputeBufferSorter<uint, uint> _sorter;
    private BVHConstructor _bvhConstructor;
    private MeshBufferContainer[] _containers;                      // one replica per GPU
    private ComputeBufferSorter<uint, uint>[] _sorters;
    private BVHConstructor[] _bvhConstructors;
    private IntPtr[] _done;                                         // rank r's "my share is in the frame buffer" event
    private IntPtr _consumed;                                       // rank 0's "the previous frame has been read" event
    private Material _imageComposerMaterial;
    private static readonly int ObjectTexture = Shader.PropertyToID("_ObjectTexture");

    private NativeBuffer _hits;            // RaycastResult {t, triangle, u, v} per pixel (Raytracing.compute:23-35)
    private NativeBuffer _image;           // RGBA16F per pixel
    private NativeBuffer _texels;          // _meshTexture as RGBA8
    private int _texW, _texH, _width, _height;
    private Texture2D _objectTexture;
    private ushort[] _imageHost;

    void Awake()
    {
        _camera = GetComponent<Camera>();

        if (_gpuDevices != null && _gpuDevices.Length > 0) LbvhContext.Devices = _gpuDevices;
        int ranks = LbvhContext.Count;
        _containers = new MeshBufferContainer[ranks];
        _sorters = new ComputeBufferSorter<uint, uint>[ranks];
        _bvhConstructors = new BVHConstructor[ranks];
        _done = new IntPtr[ranks];
        for (int r = 0; r < ranks; r++)
        {
            // the reference's Awake() chain (RaytracingMeshDrawer.cs:34-51), once per GPU: whatever is constructed while rank r is
            // current lives on rank r's device.  All calls are asynchronous: the GPUs build side by side
            LbvhContext.Current = r;

            _container = new MeshBufferContainer(_mesh);
            Debug.Log("Triangles Length " + _container.TrianglesLength);
            _sorter = new ComputeBufferSorter<uint, uint>(_container.TrianglesLength, _container.Keys, _container.TriangleIndex, _shaderContainer);
            _sorter.Sort();

            _container.DistributeKeys();

            _bvhConstructor = new BVHConstructor(_container.TrianglesLength,
                _container.Keys,
                _container.TriangleIndex,
                _container.TriangleAABB,
                _container.BvhInternalNode,
                _container.BvhLeafNode,
                _container.BvhData,
                _shaderContainer);

            _bvhConstructor.ConstructTree();
            _bvhConstructor.ConstructBVH();

            _container.GetAllGpuData();

            // derived traversal scene for LBVH_TRACE_FAST (fused 64-byte nodes over the same sorted triangles)
            IntPtr ctx = LbvhContext.Handle;
            LbvhNative.Scene scene = _container.NativeScene();
            LbvhNative.Check(ctx, LbvhNative.lbvh_build_fast_scene(ctx, ref scene, new[] { -125f, -125f, -125f }, new[] { 125f, 125f, 125f }));

            _containers[r] = _container; _sorters[r] = _sorter; _bvhConstructors[r] = _bvhConstructor;
            LbvhNative.Check(ctx, LbvhNative.lbvh_sync_event_create(ctx, out _done[r]));
            if (r != 0) LbvhNative.Check(ctx, LbvhNative.lbvh_peer_enable(ctx, LbvhContext.Devices[0]));     // rank r stores into rank 0's frame
        }
        LbvhContext.Current = 0;                                    // the frame, the texture and the image live on rank 0
        _container = _containers[0]; _sorter = _sorters[0]; _bvhConstructor = _bvhConstructors[0];
        LbvhNative.Check(LbvhContext.Handle, LbvhNative.lbvh_sync_event_create(LbvhContext.Handle, out _consumed));

        Color32[] px = ((Texture2D)_meshTexture).GetPixels32();     // row 0 = v = 0, as the sampler addresses it
        _texW = _meshTexture.width;
        _texH = _meshTexture.height;
        _texels = new NativeBuffer(px.Length, 4);
        _texels.SetData(px);

        _imageComposerMaterial = new Material(_imageComposer);
        Resize(Screen.width, Screen.height);
    }

    void Resize(int width, int height)
    {
        LbvhContext.Sync();                                         // nobody is writing the old frame buffer
        LbvhContext.Current = 0;
        _hits?.Release();
        _image?.Release();
        _width = width;
        _height = height;
        _hits = new NativeBuffer(width * height, Marshal.SizeOf(typeof(LbvhNative.Hit)));
        _image = new NativeBuffer(width * height, 8);
        _imageHost = new ushort[width * height * 4];
        _objectTexture = new Texture2D(width, height, TextureFormat.RGBAHalf, false, true);
        _imageComposerMaterial.SetTexture(ObjectTexture, _objectTexture);
    }

    private void Update()
    {
        if (Screen.width != _width || Screen.height != _height) Resize(Screen.width, Screen.height);

        Matrix4x4 m = _camera.cameraToWorldMatrix;
        var cam = new LbvhNative.Camera
        {
            screenWidth = _width, screenHeight = _height,
            cameraFov = Mathf.Tan(_camera.fieldOfView * Mathf.Deg2Rad / 2),      // RaytracingMeshDrawer.cs:80
            nearPlane = _camera.nearClipPlane,                                    // _ProjectionParams.y, Raytracing.compute:108
            m00 = m.m00, m01 = m.m01, m02 = m.m02, m03 = m.m03,
            m10 = m.m10, m11 = m.m11, m12 = m.m12, m13 = m.m13,
            m20 = m.m20, m21 = m.m21, m22 = m.m22, m23 = m.m23,
            m30 = m.m30, m31 = m.m31, m32 = m.m32, m33 = m.m33,
        };
        IntPtr ctx = LbvhContext.HandleOf(0);
        int ranks = LbvhContext.Count;
        // rank 0 has read the previous frame (its shade + read-back are enqueued before this point): only then may the others
        // overwrite the buffer — they wait for this event on the device
        LbvhNative.Check(ctx, LbvhNative.lbvh_event_record(ctx, _consumed));
        for (int r = 1; r < ranks; r++)
            LbvhNative.Check(LbvhContext.HandleOf(r), LbvhNative.lbvh_event_wait(LbvhContext.HandleOf(r), _consumed));
        for (int r = 0; r < ranks; r++)
        {
            // every GPU's share of the frame, written at its pixels of rank 0's buffer (one launch each, enqueued round-robin)
            IntPtr rc = LbvhContext.HandleOf(r);
            LbvhNative.Scene scene = _containers[r].NativeScene();
            LbvhNative.Check(rc, LbvhNative.lbvh_trace_primary_shard(rc, ref cam, (uint)r, (uint)ranks, ref scene,
                                                                     _exactTies ? LbvhNative.TRACE_FAST_EXACT : LbvhNative.TRACE_FAST,
                                                                     _hits.Pointer, IntPtr.Zero));
            if (r != 0) LbvhNative.Check(rc, LbvhNative.lbvh_event_record(rc, _done[r]));
        }
        for (int r = 1; r < ranks; r++)
            LbvhNative.Check(ctx, LbvhNative.lbvh_event_wait(ctx, _done[r]));       // the gather: a device-side wait, nothing is copied
        LbvhNative.Check(ctx, LbvhNative.lbvh_shade(ctx, _hits.Pointer, (UIntPtr)(ulong)(_width * _height), _container.TriangleData.Pointer,
                                                    _texels.Pointer, _texW, _texH, _image.Pointer));
        _image.GetData(_imageHost);                                 // blocks until the frame is done
        _objectTexture.SetPixelData(_imageHost, 0);
        _objectTexture.Apply(false);
    }

    private void OnRenderImage(RenderTexture src, RenderTexture dest)
    {
        Graphics.Blit(src, dest, _imageComposerMaterial);           // ImageComposer.shader:44-52, unchanged
    }

    private void OnDestroy()
    {
        LbvhContext.Sync();                                         // no GPU may still be storing into the frame buffer
        for (int r = 0; r < _containers.Length; r++)
        {
            _sorters[r].Dispose();
            _containers[r].Dispose();
            _bvhConstructors[r].Dispose();
            LbvhNative.lbvh_event_destroy(LbvhContext.HandleOf(r), _done[r]);
        }
        LbvhNative.lbvh_event_destroy(LbvhContext.HandleOf(0), _consumed);
        _hits?.Release();
        _image?.Release();
        _texels?.Release();
        LbvhContext.Shutdown();
    }
}
