// LbvhContext.cs — the one native context the re-hosted classes share.
//
// The reference never names its device: `new ComputeBuffer(...)` and `ComputeShader.Dispatch` use Unity's implicit
// graphics device (Assets/_Scripts/DataBuffer.cs:27, ComputeBufferSorter.cs:104-115).  The native library wants an
// explicit lbvh_context (one HIP device + stream + scratch); this static holder plays the implicit device's role so that
// the constructors of DataBuffer / MeshBufferContainer / ComputeBufferSorter / BVHConstructor keep the reference's
// signatures.  Main-thread only, like every Unity API the reference calls.
//
// SOURCE ONLY — this image has no C# toolchain (no dotnet / mono / csc / Unity); tests/test_csharp_surface.py checks the
// public surface of these files against the reference's by parsing both.
using System;

public static class LbvhContext
{
    static IntPtr _handle = IntPtr.Zero;
    static int _device = 0;

    /// GPU the context is created on (set before the first buffer is allocated; default 0).
    public static int Device
    {
        get => _device;
        set
        {
            if (_handle != IntPtr.Zero && value != _device)
                throw new InvalidOperationException("LbvhContext.Device changed after the context was created");
            _device = value;
        }
    }

    /// The lbvh_context*, created on first use (lbvh_create: LBVH_ERR_NO_DEVICE without a gfx950 GPU — there is no CPU path).
    public static IntPtr Handle
    {
        get
        {
            if (_handle == IntPtr.Zero)
            {
                int abi = LbvhNative.lbvh_abi_version();
                if (abi != LbvhNative.ABI_VERSION)
                    throw new InvalidOperationException($"liblbvh ABI {abi}, binding written for {LbvhNative.ABI_VERSION}");
                LbvhNative.Check(IntPtr.Zero, LbvhNative.lbvh_create(_device, out _handle));
            }
            return _handle;
        }
    }

    /// Blocks until everything enqueued so far has finished (the only other sync points are GetData calls).
    public static void Sync() => LbvhNative.Check(Handle, LbvhNative.lbvh_sync(Handle));

    /// Frees the library's scratch and its stream; every NativeBuffer must have been released before.
    public static void Shutdown()
    {
        if (_handle == IntPtr.Zero) return;
        LbvhNative.lbvh_destroy(_handle);
        _handle = IntPtr.Zero;
    }
}
