// NativeBuffer.cs — what UnityEngine.ComputeBuffer is to the reference, for HIP device memory.
//
// Only the members the reference touches exist: the (count, stride) constructor (Assets/_Scripts/DataBuffer.cs:27),
// SetData / GetData with a managed array (DataBuffer.cs:20,52,58; ComputeBufferSorter.cs:93,109,139), Release
// (DataBuffer.cs:74), plus `count` / `stride`.  `Pointer` is the device address the P/Invoke calls take.
// SOURCE ONLY (no C# toolchain in the build image).
using System;
using System.Runtime.InteropServices;

public sealed class NativeBuffer : IDisposable
{
    public IntPtr Pointer { get; private set; }
    public int count { get; }
    public int stride { get; }
    /// The native context (= GPU) this buffer lives on: the one that was current when it was created (LbvhContext.Current).
    /// The re-hosted classes take the context of their calls from their buffers, so a replica built while GPU k was current
    /// keeps working on GPU k whatever is current later.
    public IntPtr Context { get; }

    public NativeBuffer(int count, int stride)
    {
        if (count <= 0 || stride <= 0) throw new ArgumentException("NativeBuffer: count and stride must be positive");
        this.count = count;
        this.stride = stride;
        IntPtr ctx = LbvhContext.Handle;
        Context = ctx;
        LbvhNative.Check(ctx, LbvhNative.lbvh_buffer_alloc(ctx, (UIntPtr)(ulong)count, (UIntPtr)(ulong)stride, out IntPtr p));
        Pointer = p;
    }

    /// Bytes of a managed array handed to SetData / GetData: from the ARRAY's element type, never from this buffer's
    /// stride — the drawer reads its RGBA16F image (stride 8) back into a ushort[4 W H] (ADVICE r3: `Length * stride`
    /// priced that array at four times its size and threw on every frame).  The array may be smaller than the buffer
    /// (a prefix transfer, as ComputeBuffer allows); it may not be larger, and it must hold whole elements.
    long Bytes(Array data)
    {
        Type element = data.GetType().GetElementType();
        long bytes = element.IsPrimitive ? Buffer.ByteLength(data) : (long)data.Length * Marshal.SizeOf(element);
        if (bytes > (long)count * stride) throw new ArgumentException("NativeBuffer: array larger than the buffer");
        if (bytes % stride != 0) throw new ArgumentException("NativeBuffer: array does not hold whole buffer elements");
        return bytes;
    }

    /// Host array -> device (ComputeBuffer.SetData); blocking like Unity's.
    public void SetData(Array data)
    {
        IntPtr ctx = Context;
        GCHandle h = GCHandle.Alloc(data, GCHandleType.Pinned);
        try { LbvhNative.Check(ctx, LbvhNative.lbvh_buffer_upload(ctx, Pointer, h.AddrOfPinnedObject(), (UIntPtr)(ulong)Bytes(data))); }
        finally { h.Free(); }
    }

    /// Device -> host array (ComputeBuffer.GetData): waits for the work enqueued so far, the reference's only sync point.
    public void GetData(Array data)
    {
        IntPtr ctx = Context;
        GCHandle h = GCHandle.Alloc(data, GCHandleType.Pinned);
        try { LbvhNative.Check(ctx, LbvhNative.lbvh_buffer_download(ctx, h.AddrOfPinnedObject(), Pointer, (UIntPtr)(ulong)Bytes(data))); }
        finally { h.Free(); }
    }

    /// Every 32-bit word of the buffer = value (the fill constructors' uint.MaxValue / NullLeaf patterns), on the device.
    public void Fill(uint value)
    {
        IntPtr ctx = Context;
        LbvhNative.Check(ctx, LbvhNative.lbvh_buffer_fill_u32(ctx, Pointer, value, (UIntPtr)(ulong)((long)count * stride / 4)));
    }

    public void Release()
    {
        if (Pointer == IntPtr.Zero) return;
        LbvhNative.lbvh_buffer_free(Context, Pointer);
        Pointer = IntPtr.Zero;
    }

    public void Dispose() => Release();
}
