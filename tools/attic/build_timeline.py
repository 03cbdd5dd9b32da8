#!/usr/bin/env python3
"""Timeline of one graph-replayed rebuild (both lanes) from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python3 tools/build_timeline.py run
  python3 tools/build_timeline.py show gpurun_out/tl            (prints start / end of each kernel, us)
"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    from unitysimpleraytracing_amd import scenes
    from unitysimpleraytracing_amd.host import Context, RaytracingMeshDrawer
    tris = scenes.tiled_torus(nu=80, nv=50)
    with Context(0) as ctx:
        d = RaytracingMeshDrawer(ctx, tris).awake()
        for _ in range(8):
            d.rebuild()
            ctx.sync()
        d.on_destroy()


def run_steps():
    """rebuild + frame trace per step, as bench.py's timed loop does"""
    import ctypes as C
    from unitysimpleraytracing_amd import _native as N, layouts as L, scenes
    from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer
    W, H = 1920, 1080
    tris = scenes.tiled_torus(nu=80, nv=50)
    cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 250.0)))
    with Context(0) as ctx:
        d = RaytracingMeshDrawer(ctx, tris).awake()
        hits = DataBuffer(ctx, W * H, L.HIT)
        for _ in range(8):
            d.rebuild()
            s = d.container.scene()
            N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(cam), 0, 1, C.byref(s), L.TRACE_FAST, hits.device, None))
            ctx.sync()
        d.on_destroy()


def show(root):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"] if "Stream_Id" in r else "?"))
    rows.sort()
    # the last rebuild = everything from the last morton_aabb_kernel on
    start = max(i for i, r in enumerate(rows) if "morton_aabb" in r[2])
    t0 = rows[start][0]
    for s, e, name, q in rows[start:]:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f}  {(e - s) / 1e3:7.1f} us  q{q}  {short[:60]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "steps":
        run_steps()
    else:
        show(sys.argv[2])
