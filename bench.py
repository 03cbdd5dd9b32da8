#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "LBVH build Mtri/s + primary Mrays/s at 1080p on 1M-tri synthetic mesh".
Workload = configs[1]: 1 000 000 triangles (125 tiles of an 8 000-face bumpy torus, seed 2),
1920x1080 primary rays, camera at (0, 0, 250).

A step = one full pass of the hot path with the triangles already resident in HBM:
    Morton/AABB -> radix sort -> DistributeKeys -> Karras tree -> AABB refit (the reference's
    Awake() chain, Assets/_Scripts/RaytracingMeshDrawer.cs:30-51) -> derived fast nodes ->
    primary-ray traversal of the frame (Update(), :76-84).
Nothing is skipped or cached between steps (the tree is rebuilt from the triangles every step).
With N GPUs the BVH is replicated (every rank builds the whole tree) and the 1080p frame is
sharded across ranks in interleaved groups of 8 adjacent 8x8-pixel tiles (lbvh_trace_primary_shard,
one launch per rank); no collective touches the data path.

The JSON line carries both halves of the metric: `value` = primary Mrays/s = rays of the whole
frame / the traversal part of a step (max over ranks), `build_Mtri_s` = triangles / the build
part of a step; `ms_per_step` is the whole step (build + trace), wall clock, max over ranks.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
W, H = 1920, 1080
CAMERA_POS = (0.0, 0.0, 250.0)
N_TRIS = 1_000_000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sort-bench", action="store_true")
    ap.add_argument("--sort-keys-log2", type=int, default=26)
    ap.add_argument("--mode", choices=["fast", "reference"], default="fast")
    # cfg2 = BASELINE configs[1]/[2] (the metric's workload, default); cfg4 = configs[3]: 16 M triangles with the
    # key-range sharded sort (RCCL digit-histogram all-reduce + one all-to-all) when launched on more than one rank
    ap.add_argument("--workload", choices=["cfg2", "cfg4"], default="cfg2")
    ap.add_argument("--no-dynamic", action="store_true", help="skip the untimed cfg5 (dynamic scene + 4 bounces) extra")
    # test hooks: run the N-rank path on fewer GPUs (ranks share --device, gloo instead of RCCL)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--device", type=int, default=None)
    return ap.parse_args()


TILE_W, TILE_H, SHARD_GROUP = 8, 8, 8      # LBVH_TRACE_FAST packet size and the tile group lbvh_trace_primary_shard deals


def shard_tiles(shard_index, shard_count, width, height, tile_w=TILE_W, tile_h=TILE_H, group=SHARD_GROUP):
    """Python mirror of lbvh_trace_primary_shard's ownership rule (csrc/lbvh_trace.hip shard_tile):
    tiles of the full frame in row-major order, dealt to shards in groups of `group` adjacent tiles.
    Returns the (x0, y0, x1, y1) pixel rectangles of the shard's tiles."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    n_tiles = tiles_x * tiles_y
    out = []
    for g in range(shard_index, (n_tiles + group - 1) // group, shard_count):
        for t in range(g * group, min((g + 1) * group, n_tiles)):
            ty, tx = divmod(t, tiles_x)
            out.append((tx * tile_w, ty * tile_h, min((tx + 1) * tile_w, width), min((ty + 1) * tile_h, height)))
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    device_id = local_rank if args.device is None else args.device
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        if args.backend == "nccl":
            torch.cuda.set_device(device_id)
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", device_id))
        else:
            dist_mod.init_process_group("gloo")
        dist = dist_mod

    from unitysimpleraytracing_amd import layouts as L
    from unitysimpleraytracing_amd import scenes
    from unitysimpleraytracing_amd.host import Context, DataBuffer, RaytracingMeshDrawer

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    cfg4 = args.workload == "cfg4"
    tris = scenes.tiled_torus(nu=400, nv=160) if cfg4 else scenes.tiled_torus()      # identical on every rank (seeded)
    n_tris = len(tris)
    assert n_tris == (16 * N_TRIS if cfg4 else N_TRIS)
    cam = scenes.camera(W, H, CAMERA_POS)
    mode = L.TRACE_FAST if args.mode == "fast" else L.TRACE_REFERENCE
    sorter = None
    if cfg4 and world > 1:
        # the sharded sort mixes library kernels with RCCL collectives: one stream (torch's) orders both
        import torch
        from unitysimpleraytracing_amd.sharded_sort import ShardedSorter, sort_container
        torch.cuda.set_device(device_id)
        ctx = Context(device_id, stream=torch.cuda.current_stream().cuda_stream)
        sorter = ShardedSorter(ctx)
    else:
        ctx = Context(device_id)
    drawer = RaytracingMeshDrawer(ctx, tris)
    drawer.awake(fast=True)                            # allocates everything; untimed
    ctx.sync()

    def rebuild():
        if sorter is None:
            drawer.rebuild(fast=(mode == L.TRACE_FAST))
            return
        c = drawer.container                           # RaytracingMeshDrawer.rebuild with the sort sharded over the ranks
        c.bvh_leaf_node.fill_u32(L.NULL, mirror=False)
        c.bvh_internal_node.fill_u32(L.NULL, mirror=False)
        c.generate_keys()
        sort_container(sorter, c)
        c.distribute_keys()
        drawer.bvh_constructor.construct_tree()
        drawer.bvh_constructor.construct_bvh()
        if mode == L.TRACE_FAST:
            drawer.build_fast_scene()
    hit_buf = DataBuffer(ctx, W * H, L.HIT)             # full-frame layout on every rank
    from unitysimpleraytracing_amd import _native as N
    ccam = N.Camera.from_dict(cam)

    def trace_frame():
        # this rank's share of the frame (every world-th group of 8 adjacent 8x8-pixel tiles), one launch
        s = drawer.container.scene()
        N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(ccam), rank, world, C.byref(s), mode,
                                                           hit_buf.device, None))

    def step(ev=None):
        rebuild()
        if ev:
            ctx.record(ev[0])       # rebuild | trace
        trace_frame()
        if ev:
            ctx.record(ev[1])       # end of the step = start of the next one's rebuild

    for _ in range(args.warmup):
        step()
    ctx.sync()

    # two event records per step (each costs the stream a few microseconds): a step's rebuild starts where the
    # previous step's trace ended
    events = [(ctx.event(), ctx.event()) for _ in range(args.steps)]
    ev_start = ctx.event()
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    ctx.record(ev_start)
    for k in range(args.steps):
        step(events[k])
    ctx.sync()
    barrier()
    t1 = time.perf_counter()

    starts = [ev_start] + [e[1] for e in events[:-1]]
    build_ms = float(np.mean([ctx.elapsed_ms(s0, e[0]) for s0, e in zip(starts, events)]))
    trace_ms = float(np.mean([ctx.elapsed_ms(e[0], e[1]) for e in events]))
    wall_ms = reduce_max((t1 - t0) * 1e3 / args.steps)
    build_ms_max = reduce_max(build_ms)
    trace_ms_max = reduce_max(trace_ms)

    # ---- untimed extras (rank 0 prints them) ---------------------------------------------------
    out = None
    # every rank: hits found in its own share of the frame (one more launch of the timed trace, with counters);
    # summed over the ranks they must equal the whole frame's hit count in reference order (checked on rank 0)
    share_stats = DataBuffer(ctx, 1, L.TRACE_STATS)
    s_all = drawer.container.scene()
    N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(ccam), rank, world, C.byref(s_all), mode,
                                                       hit_buf.device, share_stats.device))
    share_hits = int(share_stats.get_data()[0]["hits"])
    if dist is not None:
        import torch
        t_h = torch.tensor([float(share_hits)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t_h)
        share_hits = int(t_h.item())
    # untimed extra for N > 1: every rank traces the WHOLE frame (N frames in flight on N GPUs) — the weak-scaling
    # counterpart of the strong-scaling `value`, reported beside it, never instead of it
    weak = None
    if dist is not None and not cfg4:
        whole = DataBuffer(ctx, W * H, L.HIT)
        s_w = drawer.container.scene()

        def whole_frame():
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s_w), mode, whole.device, None))
        for _ in range(3):
            whole_frame()
        ctx.sync()
        barrier()
        t0w = time.perf_counter()
        for _ in range(args.steps):
            whole_frame()
        ctx.sync()
        barrier()
        ms_w = reduce_max((time.perf_counter() - t0w) * 1e3 / args.steps)
        weak = {"Mrays_s": round(world * W * H / (ms_w * 1e-3) / 1e6, 2), "ms_per_frame_per_gpu": round(ms_w, 4),
                "workload": "every GPU traces the whole 1080p frame, %d frames in flight" % world}
        whole.dispose()
    sharded_sort_check = None
    if sorter is not None and rank == 0:
        # the timed steps left the sharded sort's result in the container: compare it with the one-GPU sort
        c = drawer.container
        ctx.sync()
        k_sh, i_sh = c.keys.get_data().copy(), c.triangle_index.get_data().copy()
        drawer.rebuild(fast=False)
        sharded_sort_check = bool((c.keys.get_data() == k_sh).all() and (c.triangle_index.get_data() == i_sh).all())
    if rank == 0:
        # algorithmic bytes of the traversal kernel from its own visit counters: one 64-B fused
        # node per node fetch + 48 B per triangle fetch + 16 B hit record per ray
        stats_buf = DataBuffer(ctx, 1, L.TRACE_STATS)
        full = DataBuffer(ctx, W * H, L.HIT)
        s = drawer.container.scene()
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s), mode,
                                                     full.device, stats_buf.device))
        st = stats_buf.get_data()[0]
        hit_fraction = float(st["hits"]) / (W * H)
        if mode == L.TRACE_FAST:   # packet kernel: node / triangle fetches are per 64-ray packet
            own_bytes_per_ray = (64.0 * float(st["pops"]) + 48.0 * float(st["leaf_tests"])) / (W * H) + 16.0
        else:
            own_bytes_per_ray = None
        # SURVEY.md section 8d per-ray figure in the REFERENCE's visit semantics:
        # 32 P + 24 B + 44 L + 48 T + 8, P/B/L/T from the reference-order kernel's counters
        # (tests pin them equal to the oracle's)
        N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s),
                                                     L.TRACE_REFERENCE, full.device, stats_buf.device))
        rs = stats_buf.get_data()[0]
        # guard: the fast mode (after the timed steps: cost-ordered dispatch, history) found exactly the reference's hits
        if int(rs["hits"]) != int(st["hits"]) or int(rs["hits"]) != share_hits:
            raise SystemExit(f"hit counts differ: fast {int(st['hits'])}, all shards {share_hits}, reference {int(rs['hits'])}")
        bytes_per_ray = (32.0 * float(rs["pops"]) + 24.0 * float(rs["box_hits"]) + 44.0 * float(rs["leaf_tests"])
                         + 48.0 * float(rs["tri_tests"])) / (W * H) + 8.0
        ref_counts = {k: round(float(rs[k]) / (W * H), 3) for k in ("pops", "box_hits", "leaf_tests", "tri_tests")}

        # the traversal kernel alone over the full frame, HIP events on its own stream
        reps = max(5, min(args.steps, 20))
        ctx.profile_begin()
        for _ in range(reps):
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s), mode,
                                                         full.device, None))
        prof = ctx.profile_end()
        kname = [k for k in prof if "trace_" in k][0]
        trace_kernel_ms = prof[kname][1] / prof[kname][0]
        achieved = bytes_per_ray * W * H / (trace_kernel_ms * 1e-3) / 1e9

        # per-kernel breakdown of one build
        ctx.profile_begin()
        for _ in range(5):
            drawer.rebuild(fast=(mode == L.TRACE_FAST))      # single-GPU build (no collective: the other ranks are past this)
        prof_build = {k: round(v[1] / 5.0, 4) for k, v in ctx.profile_end().items()}

        # the reference's own stages alone (a-1 .. a-8: Morton, sort, DistributeKeys, tree, refit — SURVEY 8d's
        # build_Mtri_s formula), without the derived traversal scene that `build_ms` also contains
        for _ in range(2):
            drawer.rebuild(fast=False)
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(10):
            drawer.rebuild(fast=False)
        ctx.record(e1)
        ref_build_ms = ctx.elapsed_ms(e0, e1) / 10.0
        drawer.rebuild(fast=(mode == L.TRACE_FAST))

        # measured HBM copy rate of this box (float4 copy, 1 GiB)
        nbytes = 1 << 30
        a = DataBuffer(ctx, nbytes // 4, np.uint32)
        b = DataBuffer(ctx, nbytes // 4, np.uint32)
        a.fill_u32(1)
        for _ in range(2):
            ctx.copy_probe(b.device, a.device, nbytes)
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(10):
            ctx.copy_probe(b.device, a.device, nbytes)
        ctx.record(e1)
        copy_gbs = 2.0 * nbytes * 10 / (ctx.elapsed_ms(e0, e1) * 1e-3) / 1e9
        a.dispose(); b.dispose()

        roofline = {"kernel": kname, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None if cfg4 else measured_traffic("trace_packet_kernel<false"),
                    "bytes_per_ray": round(bytes_per_ray, 1), "bytes_per_ray_basis": "reference visit order, "
                    "32P+24B+44L+48T+8 (SURVEY 8d)", "reference_visits_per_ray": ref_counts,
                    "own_bytes_per_ray": None if own_bytes_per_ray is None else round(own_bytes_per_ray, 1),
                    "own_achieved": None if own_bytes_per_ray is None else
                    round(own_bytes_per_ray * W * H / (trace_kernel_ms * 1e-3) / 1e9, 1),
                    "kernel_ms": round(trace_kernel_ms, 4),
                    "note": "achieved/frac price the kernel at the bytes of the reference's per-ray walk it replaces (SURVEY 8d); the "
                            "packet kernel itself moves own_bytes_per_ray and is bound by vector / scalar instruction issue "
                            "(instruction_issue; DESIGN.md section 7), not by HBM",
                    "measured_copy_GBs": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                    "instruction_issue": None if cfg4 else issue_counters(trace_kernel_ms)}

        sort_roofline = None
        if not args.no_sort_bench:
            sort_roofline = sort_microbench(ctx, args.sort_keys_log2, copy_gbs)

        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline and not cfg4:
            cpu_baseline = cpu_leg(tris, cam)

        dynamic = None
        if world == 1 and not args.no_dynamic and not cfg4:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import dynamic_bench                               # cfg5: animate + rebuild + primary + 4 bounces per frame
            dynamic = dynamic_bench.run(ctx, frames=10, warmup=2)

        out = {
            "metric": "LBVH build Mtri/s + primary Mrays/s at 1080p on 1M-tri synthetic mesh",
            "value": round(W * H / (trace_ms_max * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "build_Mtri_s": round(n_tris / (build_ms_max * 1e-3) / 1e6, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall_ms, 4),
            "build_ms": round(build_ms_max, 4), "trace_ms": round(trace_ms_max, 4),
            "build_reference_stages_ms": round(ref_build_ms, 4),
            "build_reference_stages_Mtri_s": round(n_tris / (ref_build_ms * 1e-3) / 1e6, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32+u32", "data": "synthetic",
            "config": {"workload": ("cfg4: 16,000,000-triangle tiled bumpy torus (400x160 quads x 125 tiles, seed 2), sort "
                                    + (f"key-range sharded over {world} GPUs (RCCL digit-histogram all-reduce + one all-to-all), "
                                       if world > 1 else "on one GPU, ") + "1920x1080 primary rays; full LBVH rebuild + frame trace per step")
                       if cfg4 else
                                   "cfg2: 1,000,000-triangle tiled bumpy torus (seed 2), 1920x1080 primary rays, "
                                   "camera (0,0,250) fov 60; full LBVH rebuild + frame trace per step",
                       "triangles": n_tris, "rays": W * H, "trace_mode": args.mode,
                       "sharding": f"rays in interleaved groups of 8 tiles over {world} GPU(s) (one launch per GPU), BVH replicated, "
                                   "no collective",
                       "hit_fraction": round(hit_fraction, 4)},
            "roofline": roofline,
            "roofline_sort_scatter": sort_roofline,
            "cpu_baseline": cpu_baseline,
            "build_kernels_ms": prof_build,
            "cfg5_dynamic": dynamic,
        }
        if sharded_sort_check is not None:
            out["sharded_sort_matches_single_gpu"] = sharded_sort_check
        if weak is not None:
            out["weak_scaling_extra"] = weak
    for e in events + [(ev_start,)]:
        for x in e:
            ctx.destroy_event(x)
    drawer.on_destroy()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


def measured_traffic(kernel_substr, largest_grid=True):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command
    (tools/prof.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per the gfx950
    note in MI355X_MICROARCH.md).  None if no profile has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*traffic.json")))
    if not files:
        return None
    table = json.load(open(files[-1]))
    rows = [(int(k.split("@")[1]), v) for k, v in table.items() if kernel_substr in k and v.get("fetch_bytes_x2") is not None
            and v.get("write_bytes") is not None]
    if not rows:
        return None
    rows.sort(key=lambda r: r[0])
    v = rows[-1][1] if largest_grid else rows[0][1]
    return {"bytes": round(v["fetch_bytes_x2"] + v["write_bytes"]), "fetch_bytes_x2": round(v["fetch_bytes_x2"]),
            "write_bytes": round(v["write_bytes"]), "source": os.path.relpath(files[-1], ROOT)}


def issue_counters(kernel_ms, clock_ghz=2.4):
    """What actually bounds the packet kernel: the committed SQ instruction counters of the same frame
    (tools/pmc_packet.sh) priced at the issue rates — one wave64 vector instruction per 4 cycles per SIMD
    (1024 SIMDs), one scalar instruction per cycle per CU (256) — against this run's live kernel duration."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*packet_counters.json")))
    if not files:
        return None
    c = json.load(open(files[-1]))
    cycles = kernel_ms * 1e-3 * clock_ghz * 1e9
    return {"valu_per_step": c["valu_per_step"], "salu_per_step": c["salu_per_step"], "steps": c["steps"],
            "lane_utilisation": c["lane_utilisation"],
            "valu_issue_frac": round(c["valu_issue_cycles_per_simd"] / cycles, 3),
            "salu_issue_frac": round(c["salu_issue_cycles_per_cu"] / cycles, 3),
            "clock_GHz": clock_ghz, "source": os.path.relpath(files[-1], ROOT)}


def sort_microbench(ctx, log2n, copy_gbs):
    """Radix-sort micro-bench on 2^log2n uniform random (key, value) pairs: large enough that the
    pairs stream from HBM, not from L2 / Infinity Cache.  Reports the scatter (downsweep) kernel:
    algorithmic 16 B per pair per launch (8 B read + 8 B written)."""
    from unitysimpleraytracing_amd import _native as N
    from unitysimpleraytracing_amd.host import DataBuffer
    n = 1 << log2n
    rng = np.random.default_rng(3)
    keys = DataBuffer(ctx, n, np.uint32)
    vals = DataBuffer(ctx, n, np.uint32)
    keys.local[:] = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    vals.local[:] = np.arange(n, dtype=np.uint32)
    reps = 3
    total_ms = 0.0
    prof_sum = {}
    for r in range(reps + 1):
        keys.sync(); vals.sync()
        e0, e1 = ctx.event(), ctx.event()
        if r > 0:
            ctx.profile_begin()
        ctx.record(e0)
        N.check(ctx.handle, N.lib.lbvh_sort_pairs(ctx.handle, keys.device, vals.device, n))
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)
        if r > 0:
            total_ms += ms
            for k, v in ctx.profile_end().items():
                a = prof_sum.setdefault(k, [0, 0.0])
                a[0] += v[0]; a[1] += v[1]
    k = keys.get_data()
    assert (k[1:] >= k[:-1]).all()
    keys.dispose(); vals.dispose()
    sort_ms = total_ms / reps
    down = [v for name, v in prof_sum.items() if "onesweep" in name][0]
    kernel_ms = down[1] / down[0]
    achieved = 16.0 * n / (kernel_ms * 1e-3) / 1e9
    return {"kernel": "sort scatter pass", "keys": n, "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic("sort_onesweep_kernel") if log2n == 26 else None,
            "kernel_ms": round(kernel_ms, 4), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
            "sort_Gkeys_s": round(n / (sort_ms * 1e-3) / 1e9, 3), "sort_ms": round(sort_ms, 3),
            "kernels_ms": {name: round(v[1] / reps, 4) for name, v in prof_sum.items()}}


def cpu_leg(tris, cam):
    """CPU baseline (kind "port"): the oracle — a C restatement of the reference's C#/HLSL, the
    only runnable form of it here (no dotnet/mono/dxc) — on this box's host cores with OpenMP.
    Bounded sample: full 1 M-triangle builds for ~5 s, then the 1080p frame subsampled on a pixel
    grid chosen from a pilot run so the traversal leg takes ~10 s."""
    import oracle as O
    threads = O.num_threads()
    cap = ((len(tris) + 1023) // 1024) * 1024
    t0 = time.perf_counter()
    builds = 0
    while True:
        b = O.Built(tris, capacity=cap, threads=threads)
        builds += 1
        if time.perf_counter() - t0 > 5.0 or builds >= 20:
            break
    build_s = (time.perf_counter() - t0) / builds
    p0 = time.perf_counter()
    pilot, _ = O.trace_primary(b, cam, step=(16, 16), threads=threads)
    rate = pilot.size / max(time.perf_counter() - p0, 1e-6)              # rays/s estimate
    step = 1
    for s_ in (1, 2, 4, 8):
        step = s_
        if (W // s_) * (H // s_) / rate <= 10.0:
            break
    t1 = time.perf_counter()
    hits, st = O.trace_primary(b, cam, step=(step, step), threads=threads)
    t2 = time.perf_counter()
    nrays = hits.size
    return {"value": round(nrays / (t2 - t1) / 1e6, 4), "unit": "Mrays/s",
            "build_Mtri_s": round(len(tris) / build_s / 1e6, 4), "cores": threads, "kind": "port",
            "sample": f"{builds} full 1M-triangle builds ({build_s:.3f} s each) + the 1080p frame sampled every "
                      f"{step} pixel(s) in x and y ({nrays} rays, {t2 - t1:.2f} s); reference visit order; OpenMP over "
                      f"triangles / internal nodes / rays, serial LSD radix sort, DistributeKeys and refit"}


if __name__ == "__main__":
    main()
