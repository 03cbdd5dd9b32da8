// ComputeBufferSorter.Native.cs — Assets/_Scripts/ComputeBufferSorter.cs re-hosted on liblbvh.so.
//
// Same generic class, constructor (dataLength, keys, values, shaderContainer), Sort() and Dispose() as the reference.
// Sort() is ONE native call: a stable LSD radix sort of the (key, value) pairs over the whole padded capacity
// (Constants.DATA_ARRAY_COUNT, as the reference's Dispatch(BLOCK_SIZE) covers it), 4 passes of 8 bits, in place in the
// caller's buffers — instead of 4 x {LocalRadixSort, PreScan, BlockSum, GlobalScan, GlobalRadixSort} dispatches with
// five blocking read-backs and an O(n) CPU validation per pass (ComputeBufferSorter.cs:100-126).  The ping-pong and
// histogram scratch the reference allocates here (:58-62) lives inside the native context.
// `shaderContainer` is accepted and ignored (there are no shaders to look up); it may be null.
// SOURCE ONLY (no C# toolchain in the build image); surface checked by tests/test_csharp_surface.py.
using System;
using UnityEngine;

public class ComputeBufferSorter<TKey, TValue> : IDisposable where TKey : struct, IComparable where TValue : struct
{
    /// The reference validates every pass on the CPU (:118-125, 150-177); here one optional sortedness check after Sort().
    public static bool ValidateAfterSort = false;

    readonly NativeBuffer _keys;
    readonly NativeBuffer _values;
    readonly uint _dataLength;

    public ComputeBufferSorter(uint dataLength, NativeBuffer keys, NativeBuffer values, IShaderContainer shaderContainer)
    {
        // the reference's GetRadix throws for anything but uint / ulong keys (:180-191); the native sort takes 32-bit pairs
        if (typeof(TKey) != typeof(uint) || typeof(TValue) != typeof(uint))
            throw new NotSupportedException("the native sort takes (uint key, uint value) pairs");
        if (keys.stride != 4 || values.stride != 4 || keys.count != values.count)
            throw new ArgumentException("keys and values must be uint buffers of equal length");
        _keys = keys;
        _values = values;
        _dataLength = dataLength;
    }

    public void Sort()
    {
        IntPtr ctx = _keys.Context;                      // the GPU the pairs live on
        LbvhNative.Check(ctx, LbvhNative.lbvh_sort_pairs(ctx, _keys.Pointer, _values.Pointer, (uint)_keys.count));
        if (ValidateAfterSort) ValidateSortedData();
    }

    void ValidateSortedData()                       // ComputeBufferSorter.cs:150-177, on the first dataLength keys
    {
        uint[] sorted = new uint[_keys.count];
        _keys.GetData(sorted);
        for (uint i = 1; i < _dataLength; i++)
            if (sorted[i] < sorted[i - 1]) { Debug.LogError("Output data has unsorted element on index " + i); return; }
        Debug.Log("Output data is sorted");
    }

    public void Dispose()
    {
        // nothing of its own to free: keys / values belong to the caller, the scratch to the native context
    }
}
