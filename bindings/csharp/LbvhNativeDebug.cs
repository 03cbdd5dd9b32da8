// LbvhNativeDebug.cs — P/Invoke table of include/lbvh_debug.h: the test hooks and measurement aids of liblbvh.so.
//
// NOT part of the drop-in: none of the re-hosted reference classes calls into this file, and a Unity project can leave it out.
// SOURCE ONLY like LbvhNative.cs (no C# toolchain in this image); tests/test_abi.py checks names and arities against the header.
using System;
using System.Runtime.InteropServices;

public static class LbvhNativeDebug
{
    const string Lib = "lbvh";

    // lbvh_debug_switch: per-context switches that were environment variables until round 4 (all 0 in the product)
    public const uint SWITCH_SORT_QUEUES = 0, SWITCH_COLD_ORDER = 1, SWITCH_BUILD_FORM = 2, SWITCH_FRAME_WAIT_MS = 3, SWITCH_SORT_FORM = 4, SWITCH_FAIL_RESERVE = 5;
    [DllImport(Lib)] public static extern int lbvh_debug_switch(IntPtr ctx, uint which, uint value);
    // host-side model of the sort's tile hand-out order (no GPU involved)
    [DllImport(Lib)] public static extern uint lbvh_debug_sort_ticket_tile(uint k, uint x, uint group, uint queues);
    [DllImport(Lib)] public static extern int lbvh_debug_ray_stack_split(IntPtr ctx, uint ldsEntries);
    [DllImport(Lib)] public static extern int lbvh_debug_ray_walker(IntPtr ctx, uint walker);
    [DllImport(Lib)] public static extern int lbvh_debug_ray_stack_limit(IntPtr ctx, uint deepEntries);
    // measurement: the four-wide per-ray walkers add {rays, node lines fetched, triangles tested} (3 x ulong) while set
    [DllImport(Lib)] public static extern int lbvh_ray_stats_target(IntPtr ctx, IntPtr dStats);
    // measurement helper: shader clock held under a vector-ALU-bound load, MHz
    [DllImport(Lib)] public static extern int lbvh_clock_probe(IntPtr ctx, out float shaderMhz);
    // one LBVH_TRACE_FAST frame that also records the node fetches of every 8x8-pixel tile
    [DllImport(Lib)] public static extern int lbvh_trace_tile_costs(IntPtr ctx, ref LbvhNative.Camera camera, ref LbvhNative.Scene scene, IntPtr dHits,
        IntPtr dStats, IntPtr dTileSteps);
    [DllImport(Lib)] public static extern int lbvh_profile_begin(IntPtr ctx);
    [DllImport(Lib)] public static extern int lbvh_profile_end(IntPtr ctx, [Out] LbvhNative.ProfileRow[] rows, int maxRows, out int nRows);
    [DllImport(Lib)] public static extern int lbvh_copy_bandwidth_probe(IntPtr ctx, IntPtr dDst, IntPtr dSrc, UIntPtr bytes);
}
