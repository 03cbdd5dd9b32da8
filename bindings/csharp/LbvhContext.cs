// LbvhContext.cs — the native contexts the re-hosted classes share: one per GPU.
//
// The reference never names its device: `new ComputeBuffer(...)` and `ComputeShader.Dispatch` use Unity's implicit
// graphics device (Assets/_Scripts/DataBuffer.cs:27, ComputeBufferSorter.cs:104-115).  The native library wants an
// explicit lbvh_context (one HIP device + stream + scratch); this static holder plays the implicit device's role so that
// the constructors of DataBuffer / MeshBufferContainer / ComputeBufferSorter / BVHConstructor keep the reference's
// signatures.  With several GPUs (`Devices`, BASELINE configs[2]: BVH replicated, rays sharded) there is one context per
// device and a CURRENT one: whatever is constructed takes the current context and keeps it (NativeBuffer.Context), so
// RaytracingMeshDrawer builds a replica of the scene per GPU by running the reference's Awake() chain once per device.
// Main-thread only, like every Unity API the reference calls.
//
// SOURCE ONLY — this image has no C# toolchain (no dotnet / mono / csc / Unity); tests/test_csharp_surface.py checks the
// public surface of these files against the reference's by parsing both.
using System;

public static class LbvhContext
{
    static int[] _devices = { 0 };
    static IntPtr[] _handles = { IntPtr.Zero };
    static int _current = 0;

    /// The GPUs to use, in rank order (rank 0 owns the frame); set before the first buffer is allocated.  Default: { 0 }.
    public static int[] Devices
    {
        get => (int[])_devices.Clone();
        set
        {
            if (value == null || value.Length == 0) throw new ArgumentException("LbvhContext.Devices: at least one device");
            foreach (IntPtr h in _handles)
                if (h != IntPtr.Zero) throw new InvalidOperationException("LbvhContext.Devices changed after a context was created");
            _devices = (int[])value.Clone();
            _handles = new IntPtr[_devices.Length];
            _current = 0;
        }
    }

    /// One GPU (the single-device form of Devices).
    public static int Device
    {
        get => _devices[0];
        set { if (_devices.Length != 1 || _devices[0] != value) Devices = new[] { value }; }
    }

    public static int Count => _devices.Length;

    /// Rank (index into Devices) whose context new buffers and re-hosted objects take.
    public static int Current
    {
        get => _current;
        set
        {
            if (value < 0 || value >= _devices.Length) throw new ArgumentOutOfRangeException(nameof(value));
            _current = value;
        }
    }

    /// The current rank's lbvh_context*, created on first use (lbvh_create: LBVH_ERR_NO_DEVICE without a gfx950 GPU — there is no CPU path).
    public static IntPtr Handle => HandleOf(_current);

    public static IntPtr HandleOf(int rank)
    {
        if (_handles[rank] == IntPtr.Zero)
        {
            int abi = LbvhNative.lbvh_abi_version();
            if (abi != LbvhNative.ABI_VERSION)
                throw new InvalidOperationException($"liblbvh ABI {abi}, binding written for {LbvhNative.ABI_VERSION}");
            LbvhNative.Check(IntPtr.Zero, LbvhNative.lbvh_create(_devices[rank], out _handles[rank]));
        }
        return _handles[rank];
    }

    /// Blocks until everything enqueued so far on every GPU has finished (the only other sync points are GetData calls).
    public static void Sync()
    {
        for (int r = 0; r < _handles.Length; r++)
            if (_handles[r] != IntPtr.Zero) LbvhNative.Check(_handles[r], LbvhNative.lbvh_sync(_handles[r]));
    }

    /// Frees the library's scratch and its streams; every NativeBuffer must have been released before.
    public static void Shutdown()
    {
        for (int r = 0; r < _handles.Length; r++)
        {
            if (_handles[r] == IntPtr.Zero) continue;
            LbvhNative.lbvh_destroy(_handles[r]);
            _handles[r] = IntPtr.Zero;
        }
    }
}
