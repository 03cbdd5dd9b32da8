// DataBuffer.Native.cs — the reference's host/device buffer pair (Assets/_Scripts/DataBuffer.cs) over HIP memory.
//
// Public surface = the reference's: DeviceBuffer, LocalBuffer, DataBuffer(size), DataBuffer(size, initialValue), the
// uint indexer that reads back lazily, GetData, Sync, ToString, Dispose — with NativeBuffer where the reference says
// ComputeBuffer.  It takes the place of the reference file in a Unity project (same class name).
// SOURCE ONLY (no C# toolchain in the build image); surface checked by tests/test_csharp_surface.py.
using System;
using System.Runtime.InteropServices;

public class DataBuffer<T> : IDisposable where T : struct
{
    readonly NativeBuffer _device;
    readonly T[] _host;
    bool _hostIsCurrent;          // false while the device may hold newer data than _host (or _host has unsent edits)

    public NativeBuffer DeviceBuffer => _device;
    public T[] LocalBuffer => _host;

    public DataBuffer(int size)                                   // DataBuffer.cs:25-30: device memory + an unsynced mirror
    {
        _device = new NativeBuffer(size, Marshal.SizeOf(typeof(T)));
        _host = new T[size];
    }

    public DataBuffer(int size, T initialValue) : this(size)      // DataBuffer.cs:14-23: every element = initialValue, both sides
    {
        Array.Fill(_host, initialValue);
        Sync();
    }

    public T this[uint i]                                         // DataBuffer.cs:32-48
    {
        get { if (!_hostIsCurrent) GetData(); return _host[i]; }
        set { _host[i] = value; _hostIsCurrent = false; }
    }

    public void GetData() { _device.GetData(_host); _hostIsCurrent = true; }     // :50-54, blocks (the reference's sync point)

    public void Sync() { _device.SetData(_host); _hostIsCurrent = true; }        // :56-60

    public override string ToString()                                            // :62-70
    {
        if (!_hostIsCurrent) GetData();
        return Utils.ArrayToString(_host).ToString();
    }

    public void Dispose() => _device.Release();                                  // :72-75
}
