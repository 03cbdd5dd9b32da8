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

`roofline` describes the dominant kernel (the packet traversal) against HBM, SURVEY 8(d)'s roofline for every stage:
  achieved = the kernel's OWN algorithmic bytes per launch (64 B per node line + 64 B per triangle line fetched per
      64-ray packet + 16 B hit record per ray, from this run's visit counters) / its live HIP-event duration; peak 8 TB/s;
      frac = achieved / peak — a small number: the walk prunes and the scene it walks is cache-resident;
  roofline.traffic: HBM-side bytes per launch from live FETCH_SIZE / WRITE_SIZE child passes (or null — never replayed
      from a committed file);
  roofline.valu_issue: the busiest unit of the kernel — wave64 vector instructions per second (SQ_INSTS_VALU of a live
      rocprofv3 child pass of this very run) against 1024 SIMDs x the measured shader clock / 2 cycles (the guide's
      v_fma_f32 rate with other waves resident), and the share of SIMD cycles with a vector instruction executing.  Not
      "the" bound: with 17 % of these instructions moved to the matrix pipe the frame is 0 - 2 % shorter (DESIGN 13.3,
      profiles/r4/d_*): a step is a dependent chain across the vector pipe, the scalar unit and the memory system.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 1024                # 256 CUs x 4 SIMDs
# /opt/skills/guides/MI355X_MICROARCH.md, 'Per-instruction cycle constants': a wave64 v_fma_f32 issues in 2 cycles on the
# SIMD-32 when other waves are resident (4 is what ONE wave alone sustains).  tools/ubench/instcost.hip on this chip
# (profiles/r3/b_instcost.txt): v_mul / v_add / v_fma / v_mov 2.3 cycles at >= 4 waves per SIMD; every v_cmp, v_min / v_max /
# min3 / max3, anything with a DPP or SGPR operand, v_readlane, v_cndmask (VOP3) 4.3; VOP2 v_cndmask 23; one wave alone 5.1 - 6.3
VALU_ISSUE_CYCLES = 2
W, H = 1920, 1080
CAMERA_POS = (0.0, 0.0, 250.0)
N_TRIS = 1_000_000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sort-bench", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true", help="skip the rocprofv3 child passes (traffic = null)")
    ap.add_argument("--sort-keys-log2", type=int, default=26)
    ap.add_argument("--mode", choices=["fast", "reference"], default="fast")
    # cfg2 = BASELINE configs[1]/[2] (the metric's workload, default); cfg4 = configs[3]: 16 M triangles with the
    # key-range sharded sort (RCCL digit-histogram all-reduce + one all-to-all) when launched on more than one rank
    ap.add_argument("--workload", choices=["cfg2", "cfg4"], default="cfg2")
    ap.add_argument("--no-dynamic", action="store_true", help="skip the untimed cfg5 (dynamic scene + 4 bounces) extra")
    ap.add_argument("--no-in-flight", action="store_true", help="skip the untimed steps-in-flight extra (2 / 3 contexts on one GPU)")
    # test hooks: run the N-rank path on fewer GPUs (ranks share --device, gloo instead of RCCL)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--device", type=int, default=None)
    # N > 1: how every rank's hit records reach rank 0's full-frame buffer inside the timed step (frame_gather.py):
    # peer = traced straight into rank 0's memory (IPC-mapped, xGMI stores) + device-side completion flags;
    # packed = contiguous shares + one torch.distributed gather + lbvh_frame_unpack; auto = peer if its self-test passes
    ap.add_argument("--gather", choices=["auto", "peer", "packed"], default="auto")
    # --gpus N from a plain launch: the parent ends every rank and exits 124 when the whole run takes longer than this (a
    # collective library that never connects must not hold the caller until ITS timeout)
    ap.add_argument("--launch-deadline", type=float, default=900.0, metavar="SECONDS")
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


def yawed(cam, yaw_deg):
    """the camera dict turned about the world Y axis (position kept): a moving-camera frame"""
    cy, sy = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    yaw = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    m = np.array(cam["camera_to_world"], dtype=np.float32).reshape(4, 4).copy()
    m[:3, :3] = (yaw @ m[:3, :3].astype(np.float64)).astype(np.float32)
    out = dict(cam)
    out["camera_to_world"] = m.reshape(-1).copy()
    return out


def self_launch(n, deadline_s):
    """`python bench.py --gpus N` from a plain launch: start the N ranks as CHILD processes (one per GPU, the environment
    torch.distributed.run would give them) and exit with their worst return code.  Nothing in this parent has touched HIP
    or torch at this point (no exec of a GPU-initialised process anywhere); rank 0's JSON line goes straight to stdout.
    The whole launch has a deadline (--launch-deadline, 900 s): ranks that hang in the rendezvous, in RCCL's set-up or in a
    collective are ended — SIGTERM, five seconds later SIGKILL; children are fresh processes, nothing is exec'ed — and the parent
    exits 124 having said which ranks were still running."""
    import subprocess
    import tempfile
    # rendezvous through a file the ranks share (torch.distributed's file:// store), not through a TCP port picked here and
    # released before rank 0 binds it (another process could take it in between)
    rdv_dir = tempfile.mkdtemp(prefix="lbvh_bench_rdv_")
    procs = []
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC — without it RCCL's own buffer sharing AND
    # the peer-mapped frame gather fail with "hipIpcGetMemHandle: invalid argument" (the task environment exports it already; a
    # caller's own value is never overridden).  Said once on stderr so that a first contact with N GPUs shows what was in force.
    ipc_mode = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    print(f"[bench] launching {n} ranks; HSA_ENABLE_IPC_MODE_LEGACY={'0 (set here)' if ipc_mode is None else ipc_mode + ' (inherited)'}; "
          f"overall deadline {deadline_s:.0f} s", file=sys.stderr, flush=True)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   LBVH_BENCH_RENDEZVOUS_FILE=os.path.join(rdv_dir, "store"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # stdout carries ONE JSON line, rank 0's: whatever another rank prints goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    worst = 0
    overall = time.time() + deadline_s
    try:
        pending = list(procs)
        deadline = None          # set when a rank has failed: the others get ten seconds to say why themselves
        kill_at = None           # set when the pending ranks have been sent SIGTERM
        while pending:
            for p_ in list(pending):
                rc = p_.poll()
                if rc is None:
                    continue
                pending.remove(p_)
                if rc != 0:
                    worst = worst or rc
                    # one rank failed: the others would wait in a collective for ever — but a rank that fails for a reason of its
                    # own (no device for it) says so itself if it is given the time: ten seconds' grace, then they are ended
                    deadline = deadline or time.time() + 10.0
            now = time.time()
            if pending and kill_at is None and now > overall:
                print(f"[bench] the launch did not finish within {deadline_s:.0f} s: ending rank(s) "
                      f"{[procs.index(q) for q in pending]} (hung in the rendezvous, the collective library's set-up or a collective)",
                      file=sys.stderr, flush=True)
                worst = 124
                deadline = now - 1.0
            if pending and kill_at is None and deadline is not None and now > deadline:
                for q in pending:
                    q.terminate()
                kill_at = now + 5.0
            if pending and kill_at is not None and now > kill_at:
                for q in pending:
                    q.kill()
                kill_at = now + 3600.0
            time.sleep(0.05)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
        import shutil
        shutil.rmtree(rdv_dir, ignore_errors=True)
    raise SystemExit(worst)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, args.launch_deadline)
    if os.environ.get("LBVH_BENCH_TEST_HANG") == "1" and "WORLD_SIZE" in os.environ:
        # test hook (tests/test_sharding.py): a rank that never gets anywhere — before HIP, torch or a rendezvous is touched
        import signal
        signal.signal(signal.SIGTERM, signal.SIG_IGN if os.environ.get("RANK") == "1" else signal.SIG_DFL)
        while True:
            time.sleep(1.0)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    dist = None
    device_id = local_rank if args.device is None else args.device
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        # The set-up must fail LOUDLY and at once, not inside a collective: count the devices before anything initialises one
        # (torch.cuda.device_count() does not create a HIP context) and say what to run instead
        visible = torch.cuda.device_count()
        if args.backend == "nccl" and (device_id < 0 or device_id >= visible):      # (gloo: lbvh_create itself says "no HIP device")
            raise SystemExit(f"[bench] rank {rank}/{world}: needs HIP device {device_id}, this process sees {visible} "
                             f"(--backend {args.backend}); one rank per GPU: --gpus N <= the node's GPUs, or every rank on one GPU: "
                             "--backend gloo --device 0")
        if args.backend == "nccl" and args.device is not None and world > 1:
            raise SystemExit("[bench] --backend nccl (RCCL) needs one GPU per rank: --device puts every rank on the same one; "
                             "use --backend gloo --device D for the N-rank code path on one GPU")
        if rank == 0:
            print(f"[bench] {world} ranks, backend {args.backend}, rank r on HIP device "
                  f"{'r' if args.device is None else args.device} of {visible} visible, rendezvous "
                  f"{'file store (self-launched)' if os.environ.get('LBVH_BENCH_RENDEZVOUS_FILE') else 'MASTER_ADDR:PORT (torchrun)'}",
                  file=sys.stderr, flush=True)
        # stdout carries ONE JSON line: whatever the communication libraries print while they connect goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            rdv = os.environ.get("LBVH_BENCH_RENDEZVOUS_FILE")          # set by self_launch; torchrun gives MASTER_ADDR / PORT
            how = dict(init_method="file://" + rdv, rank=rank, world_size=world) if rdv else {}
            if args.backend == "nccl":
                torch.cuda.set_device(device_id)
                dist_mod.init_process_group("nccl", device_id=torch.device("cuda", device_id), **how)
            else:
                dist_mod.init_process_group("gloo", **how)
            dist_mod.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
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
    # N > 1: the step ends with ONE whole frame in rank 0's buffer (the reference renders one image per Update(),
    # RaytracingMeshDrawer.cs:76-89): every rank's records travel to rank 0 inside the timed region
    gather = None
    if dist is not None:
        from unitysimpleraytracing_amd.frame_gather import FrameGather
        if args.backend == "nccl":
            import torch
            torch.cuda.set_device(device_id)
        gather = FrameGather(ctx, dist, rank, world, W, H, device_id, staged=(args.backend == "gloo"), mode=args.gather)
        if rank == 0:           # which transport `--gather auto` settled on, and why not the other (stderr: stdout is the JSON line)
            print(f"[bench] frame gather: asked for {args.gather!r}, using {gather.mode!r}"
                  + (f" (peer-mapped frame buffer unavailable: {gather.peer_error})" if gather.peer_error else ""), file=sys.stderr, flush=True)

    def trace_share(camera=None, stats=None):
        # this rank's share of the frame (every world-th group of 8 adjacent 8x8-pixel tiles), one launch, into this
        # rank's own buffer: the untimed extras (what a share costs alone)
        s = drawer.container.scene()
        N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(camera if camera is not None else ccam), rank, world,
                                                           C.byref(s), mode, hit_buf.device, stats))

    def step(ev=None):
        rebuild()
        if ev:
            ctx.record(ev[0])       # rebuild | trace
        if gather is None:
            trace_share()
        else:                       # the share + its way into rank 0's frame (+ on rank 0: the wait for everybody's)
            gather.trace_share(ccam, drawer.container.scene(), mode, own_done_event=ev[2] if ev else None,
                               own_start_event=ev[3] if ev else None)
        if ev:
            ctx.record(ev[1])       # end of the step = start of the next one's rebuild

    for _ in range(args.warmup):
        step()
    ctx.sync()

    # ---- K steps INSTRUMENTED: two event records per step split a step into rebuild | trace (a step's rebuild starts where the
    # previous step's trace ended) -> build_ms / trace_ms, hence `value` and build_Mtri_s; wall clock: `ms_per_step_instrumented`.
    # This pass runs FIRST: the timed region below is then the second run of the same K steps, on a chip whose clocks have settled
    # (with W = 3 and K = 20 the timed steps were the first 8 ms of work after seconds of host-side set-up: 0.397 ms per step
    # against 0.384 for the instrumented pass right behind them and 0.375 for the best of three such passes).
    events = [(ctx.event(), ctx.event()) + ((ctx.event(), ctx.event()) if gather is not None else ()) for _ in range(args.steps)]
    ev_start = ctx.event()
    barrier()
    ctx.sync()
    t0i = time.perf_counter()
    ctx.record(ev_start)
    for k in range(args.steps):
        step(events[k])
    ctx.sync()
    barrier()
    t1i = time.perf_counter()
    wall_instrumented_ms = reduce_max((t1i - t0i) * 1e3 / args.steps)
    # ---- the TIMED region: exactly K steps, nothing but the steps on the stream --------------------------------------------------
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    ctx.sync()
    barrier()
    t1 = time.perf_counter()

    starts = [ev_start] + [e[1] for e in events[:-1]]
    build_ms = float(np.mean([ctx.elapsed_ms(s0, e[0]) for s0, e in zip(starts, events)]))
    trace_ms = float(np.mean([ctx.elapsed_ms(e[0], e[1]) for e in events]))
    wall_ms = reduce_max((t1 - t0) * 1e3 / args.steps)
    build_ms_max = reduce_max(build_ms)
    trace_ms_max = reduce_max(trace_ms)
    # N > 1: trace_ms contains the records' way to rank 0 (on rank 0: the wait for every rank's share); the traversal alone:
    # (from behind the wait for rank 0's "frame read" word to the end of this rank's own kernel: no cross-rank skew in it)
    own_trace_ms_max = reduce_max(float(np.mean([ctx.elapsed_ms(e[3], e[2]) for e in events]))) if gather is not None else trace_ms_max

    # ---- the arrays the timed steps left behind, word for word against an independent chain (untimed) ------
    # The timed rebuild is ONE lbvh_build_scene call: two-level sort, three merged launches, the search-free tree kernel that
    # takes its boxes from the range hierarchy, all replayed from a captured graph.  The check rebuilds the same triangles with
    # the reference's stage calls one by one (MeshBufferContainer / ComputeBufferSorter / BVHConstructor mirrors: other kernels
    # throughout — four 8-bit LSD passes forced, the three-kernel DistributeKeys, the topology-only tree kernel, the
    # bottom-up refit climb) and refuses the line on any difference.
    timed_path_check = None
    if sorter is None:
        c = drawer.container
        n_ = c.triangles_length

        def snapshot():
            ctx.sync()
            return (c.keys.get_data().copy(), c.triangle_index.get_data().copy(),
                    c.bvh_internal_node.get_data().view(np.uint32).copy(), c.bvh_leaf_node.get_data().view(np.uint32).copy(),
                    c.bvh_data.get_data()["min"][: n_ - 1].copy(), c.bvh_data.get_data()["max"][: n_ - 1].copy())
        timed = snapshot()
        for buf, word in ((c.keys, 0x0BADBEEF), (c.triangle_index, 0x0BADBEEF), (c.bvh_internal_node, 0x01357246),
                          (c.bvh_leaf_node, 0x02468135), (c.bvh_data, 0x7FC00002)):
            buf.fill_u32(word, mirror=False)
        ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 1)
        drawer.rebuild(fast=False, staged=True)
        staged = snapshot()
        ctx.debug_switch(N.DEBUG_SWITCH_SORT_FORM, 0)
        names = ("keys", "sorted_indices", "internal_nodes", "leaf_nodes", "bvh_min", "bvh_max")
        differing = {nm: int(np.count_nonzero(a != b)) for nm, a, b in zip(names, timed, staged)}
        timed_path_check = {"arrays_equal_staged_four_pass_chain": not any(differing.values()), "differing_words": differing,
                            "note": "container arrays after the K timed steps vs. a staged rebuild of the same triangles "
                                    "(stage calls one by one, four-pass sort forced, separate tree + refit kernels)"}
        if any(differing.values()):
            raise SystemExit(f"[bench] rank {rank}: the timed rebuild path's arrays differ from the staged chain's: {differing}")
        for _ in range(2):
            rebuild()                   # the timed path's state again (and the sort's form hint), for the extras below
        ctx.sync()

    # ---- N > 1: the assembled frame, word for word (untimed) ------------------------------------------
    frame_check = None
    if gather is not None:
        # a fresh frame into a POISONED buffer: every record must arrive in this very frame
        if rank == 0:
            gather.frame.fill_u32(0x7FC00000, mirror=False)
        ctx.sync()
        barrier()
        gather.trace_share(ccam, drawer.container.scene(), mode)
        ctx.sync()
        barrier()
        if rank == 0:
            assembled = gather.frame.get_data().copy()
            s_c = drawer.container.scene()
            alone = DataBuffer(ctx, W * H, L.HIT)
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s_c), mode, alone.device, None))
            one_gpu = alone.get_data().copy()
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s_c), L.TRACE_REFERENCE, alone.device, None))
            ref = alone.get_data().copy()
            alone.dispose()
            words_equal = bool((assembled.view(np.uint32) == one_gpu.view(np.uint32)).all())
            t_equal = bool((assembled["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
            tie_pixels = int(np.count_nonzero(assembled["tri"] != ref["tri"]))
            frame_check = {"transport": gather.mode, "peer_unavailable_because": gather.peer_error,
                           "assembled_equals_one_gpu_frame_word_for_word": words_equal,
                           "t_equals_reference_mode_bit_for_bit": t_equal,
                           "pixels_where_the_triangle_differs_from_reference_mode": tie_pixels,
                           "note": "a fresh frame assembled in rank 0's NaN-poisoned buffer from every rank's share, compared with the "
                                   "frame rank 0 traces alone (same mode: every word) and with LBVH_TRACE_REFERENCE (t: every bit; the "
                                   "triangle differs only where two are hit at exactly the same t — DESIGN 9)"}
            if not (words_equal and t_equal):
                raise SystemExit(f"the assembled {world}-GPU frame differs from the one-GPU frame: {frame_check}")

    # ---- untimed extras ---------------------------------------------------------------------------
    tiles_x, tiles_y = (W + TILE_W - 1) // TILE_W, (H + TILE_H - 1) // TILE_H
    frame_costs = DataBuffer(ctx, tiles_x * tiles_y, np.uint32) if dist is not None else None

    def exchange_costs():
        """every rank's per-tile step counts of the frame just traced, merged (all-reduce MAX of zero-filled arrays, 130 KB)
        and handed back: under a moving camera the place a tile came from mostly belongs to another rank"""
        import torch
        from unitysimpleraytracing_amd.sharded_sort import DeviceArray
        frame_costs.fill_u32(0, mirror=False)
        ctx.trace_costs_export(frame_costs, tiles_x, tiles_y)
        ctx.sync()
        t = DeviceArray(frame_costs.device.value, tiles_x * tiles_y).tensor(torch.device("cuda", device_id))
        if args.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t.copy_(h)
            torch.cuda.synchronize()
        ctx.trace_costs_import(frame_costs, tiles_x, tiles_y)

    def timed_frames(frames, before=None, camera_of=None, after=None):
        """mean device ms of this rank's share of `frames` frames (HIP events around the trace alone), max over ranks"""
        e0, e1 = ctx.event(), ctx.event()
        total = 0.0
        for k in range(frames):
            if before is not None:
                before()
            c = None if camera_of is None else N.Camera.from_dict(camera_of(k))
            ctx.record(e0)
            trace_share(c)
            ctx.record(e1)
            total += ctx.elapsed_ms(e0, e1)
            if after is not None:
                after()
        ctx.destroy_event(e0); ctx.destroy_event(e1)
        return reduce_max(total / frames)

    reps = max(5, min(args.steps, 20))
    extras = {}
    if mode == L.TRACE_FAST:
        # the timed steps trace a static camera: every frame's dispatch order comes from the previous frame's per-tile
        # step counts.  What the same share costs (a) as a FIRST frame (history dropped before each), (b) with the
        # camera turning 1 degree per frame (the history is one frame stale), (c) without the rebuild in between
        trace_share(); ctx.sync()
        extras["trace_static_scene_ms"] = round(timed_frames(reps), 4)
        extras["trace_cold_ms"] = round(timed_frames(reps, before=ctx.trace_forget), 4)
        extras["trace_moving_ms"] = round(timed_frames(reps, camera_of=lambda k: yawed(cam, 1.0 * (k + 1))), 4)
        if dist is not None:       # the same with the ranks' per-tile costs exchanged after every frame (lbvh_trace_costs_export / _import)
            trace_share(); ctx.sync(); exchange_costs()
            extras["trace_moving_costs_exchanged_ms"] = round(timed_frames(reps, camera_of=lambda k: yawed(cam, 1.0 * (k + 1)), after=exchange_costs), 4)
        trace_share(); ctx.sync()                      # back to the timed camera's history

    out = None
    # every rank: hits found in its own share of the frame (one more launch of the timed trace, with counters);
    # summed over the ranks they must equal the whole frame's hit count in reference order (checked on rank 0)
    share_stats = DataBuffer(ctx, 1, L.TRACE_STATS)
    trace_share(stats=share_stats.device)
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
    # untimed extra: the same scene at 3840x2160 — every rank its share of ONE 4K frame, history-driven like the
    # timed frames.  At 8.3 M rays a 1/8 share is a whole 1080p frame's worth of work: this curve separates the latency floor of a
    # share (launch, filing, the heaviest tile's chain: what bounds the 1080p shares) from throughput.  Never the headline.
    frame_4k = None
    if mode == L.TRACE_FAST and not cfg4:
        W4, H4 = 3840, 2160
        cam4 = N.Camera.from_dict(scenes.camera(W4, H4, CAMERA_POS))
        hits4 = DataBuffer(ctx, W4 * H4, L.HIT)
        s4 = drawer.container.scene()

        def share_4k():
            N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(cam4), rank, world, C.byref(s4), mode, hits4.device, None))
        for _ in range(3):
            share_4k()
        e0, e1 = ctx.event(), ctx.event()
        ctx.sync()
        barrier()
        total = 0.0
        for _ in range(reps):
            ctx.record(e0)
            share_4k()
            ctx.record(e1)
            total += ctx.elapsed_ms(e0, e1)
        ctx.destroy_event(e0); ctx.destroy_event(e1)
        ms4 = reduce_max(total / reps)
        frame_4k = {"Mrays_s": round(W4 * H4 / (ms4 * 1e-3) / 1e6, 2), "share_ms": round(ms4, 4), "rays": W4 * H4,
                    "workload": "the cfg2 scene at 3840x2160: every rank its share of one frame (static camera, HIP events around the "
                                "share's launch, max over ranks; no gather, no rebuild) — an extra beside the 1080p metric"}
        hits4.dispose()
        trace_share(); ctx.sync()                      # the 1080p layout's history back for what follows
    # untimed extra, one GPU only: STEPS IN FLIGHT.  The timed step is one rebuild followed by its own trace on one stream — a
    # latency-bound half and a vector-issue-bound half that never overlap.  An application that renders a stream of frames can
    # keep two or three steps in flight on one GPU: contexts (own scene buffers, scratch, stream) taking the steps in turn, every
    # step still a full rebuild followed by the trace of THAT rebuild's scene, nothing reused across steps.  Wall clock over
    # args.steps steps, best of 3.  Reported beside `value`, never instead of it; needs no library change.
    in_flight = None
    if dist is None and mode == L.TRACE_FAST and not cfg4 and not args.no_in_flight:
        more = [Context(device_id) for _ in range(2)]
        more_drawers = [RaytracingMeshDrawer(c, tris).awake(fast=True) for c in more]
        more_hits = [DataBuffer(c, W * H, L.HIT) for c in more]
        lanes = [(ctx, drawer, hit_buf)] + list(zip(more, more_drawers, more_hits))

        def lane_step(lane):
            c, d, h = lane
            d.rebuild(fast=True)
            s_l = d.container.scene()
            N.check(c.handle, N.lib.lbvh_trace_primary(c.handle, C.byref(ccam), 0, 0, W, H, C.byref(s_l), mode, h.device, None))
        in_flight = {"workload": "cfg2 steps (full rebuild + 1080p frame of that rebuild) taken in turn by 1 / 2 / 3 contexts on one GPU; "
                                 "wall clock per step, best of 3 x %d steps" % args.steps}
        for n_l in (1, 2, 3):
            for k in range(3 * n_l):
                lane_step(lanes[k % n_l])
            for c, _, _ in lanes:
                c.sync()
            best = float("inf")
            for _ in range(3):
                t0f = time.perf_counter()
                for k in range(args.steps):
                    lane_step(lanes[k % n_l])
                for c, _, _ in lanes[:n_l]:
                    c.sync()
                best = min(best, (time.perf_counter() - t0f) * 1e3 / args.steps)
            in_flight["ms_per_step_%d" % n_l] = round(best, 4)
        same = all(int((h.get_data()["t"] < 2.0e9).sum()) == int((hit_buf.get_data()["t"] < 2.0e9).sum()) for h in more_hits)
        in_flight["frames_equal"] = bool(same and all((h.get_data() == hit_buf.get_data()).all() for h in more_hits))
        for h in more_hits:
            h.dispose()
        for d in more_drawers:
            d.on_destroy()
        for c in more:
            c.close()
        trace_share(); ctx.sync()
    sharded_sort_check = None
    if sorter is not None and rank == 0:
        # the timed steps left the sharded sort's result in the container: compare it with the one-GPU sort
        c = drawer.container
        ctx.sync()
        k_sh, i_sh = c.keys.get_data().copy(), c.triangle_index.get_data().copy()
        drawer.rebuild(fast=False)
        sharded_sort_check = bool((c.keys.get_data() == k_sh).all() and (c.triangle_index.get_data() == i_sh).all())
        drawer.rebuild(fast=(mode == L.TRACE_FAST))
    if rank == 0:
        stats_buf = DataBuffer(ctx, 1, L.TRACE_STATS)
        full = DataBuffer(ctx, W * H, L.HIT)
        s = drawer.container.scene()

        def frame_stats(camera, m):
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(camera), 0, 0, W, H, C.byref(s), m, full.device, stats_buf.device))
            return stats_buf.get_data()[0].copy()

        def own_bytes(st):          # the packet kernel's own algorithmic bytes per ray: node and triangle lines are per packet
            return (64.0 * float(st["pops"]) + 64.0 * float(st["leaf_tests"])) / (W * H) + 16.0

        def ref_bytes(rs):          # the reference algorithm's bytes per ray, 32 P + 24 B + 44 L + 48 T + 8
            return (32.0 * float(rs["pops"]) + 24.0 * float(rs["box_hits"]) + 44.0 * float(rs["leaf_tests"])
                    + 48.0 * float(rs["tri_tests"])) / (W * H) + 8.0

        st = frame_stats(ccam, mode)
        hit_fraction = float(st["hits"]) / (W * H)
        rs = frame_stats(ccam, L.TRACE_REFERENCE)
        # guard: the fast mode (after the timed steps: cost-ordered dispatch, history) found exactly the reference's hits
        if int(rs["hits"]) != int(st["hits"]) or int(rs["hits"]) != share_hits:
            raise SystemExit(f"hit counts differ: fast {int(st['hits'])}, all shards {share_hits}, reference {int(rs['hits'])}")
        ref_counts = {k: round(float(rs[k]) / (W * H), 3) for k in ("pops", "box_hits", "leaf_tests", "tri_tests")}

        def kernel_ms_of(camera):
            """the traversal kernel alone over the full frame: per-kernel HIP events on the library's own stream"""
            for _ in range(2):
                N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(camera), 0, 0, W, H, C.byref(s), mode, full.device, None))
            ctx.profile_begin()
            for _ in range(reps):
                N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(camera), 0, 0, W, H, C.byref(s), mode, full.device, None))
            prof = ctx.profile_end()
            kname = [k for k in prof if "trace_" in k][0]
            frame_ms = sum(v[1] for v in prof.values()) / reps          # traversal + tile filing
            return kname, prof[kname][1] / prof[kname][0], frame_ms

        kname, trace_kernel_ms, _ = kernel_ms_of(ccam)

        # a second camera: closer to the mesh, most rays hit (misses are cheap, so Mrays/s goes with the hit fraction)
        near_cam = N.Camera.from_dict(scenes.camera(W, H, (0.0, 0.0, 160.0)))
        st2 = frame_stats(near_cam, mode)
        rs2 = frame_stats(near_cam, L.TRACE_REFERENCE)
        if int(rs2["hits"]) != int(st2["hits"]):
            raise SystemExit(f"hit counts differ at camera z=160: fast {int(st2['hits'])}, reference {int(rs2['hits'])}")
        _, near_kernel_ms, near_frame_ms = kernel_ms_of(near_cam)
        near = {"workload": "same scene, camera (0,0,160): most rays hit", "hit_fraction": round(float(st2["hits"]) / (W * H), 4),
                "trace_ms": round(near_frame_ms, 4), "Mrays_s": round(W * H / (near_frame_ms * 1e-3) / 1e6, 2),
                "kernel_ms": round(near_kernel_ms, 4), "own_bytes_per_ray": round(own_bytes(st2), 1),
                "steps_per_packet": round(float(st2["pops"]) / (W * H / 64.0), 1)}
        kernel_ms_of(ccam)                              # history back to the timed camera

        # LBVH_TRACE_FAST_EXACT (not the timed mode): the packet walk + the reference's choice wherever two triangles are hit at
        # exactly the same t — every word of the frame equals LBVH_TRACE_REFERENCE's; what it costs, and the check itself
        exact_mode = None
        t_mismatch = 0
        if mode == L.TRACE_FAST:
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s), L.TRACE_REFERENCE, full.device, None))
            ref_frame = full.get_data().view(np.uint32).copy()
            N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s), L.TRACE_FAST, full.device, None))
            fast_frame = full.get_data().view(np.uint32).copy()
            fast_differs = int((fast_frame != ref_frame).reshape(-1, 4).any(axis=1).sum())
            # hit distances of the timed mode against LBVH_TRACE_REFERENCE on the timed frame, bit for bit: a pixel whose reference
            # winner lies in front of its own triangle's box (the one case in which a pruned walk differs) would
            # show up here; the line is refused if there is one
            t_word = L.HIT.fields["t"][1] // 4
            t_mismatch = int(np.count_nonzero(fast_frame.reshape(-1, 4)[:, t_word] != ref_frame.reshape(-1, 4)[:, t_word]))
            if t_mismatch:
                raise SystemExit(f"[bench] {t_mismatch} pixels of the timed frame differ in t between LBVH_TRACE_FAST and LBVH_TRACE_REFERENCE")
            for _ in range(3):
                N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s), L.TRACE_FAST_EXACT, full.device, None))
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            for _ in range(reps):
                N.check(ctx.handle, N.lib.lbvh_trace_primary(ctx.handle, C.byref(ccam), 0, 0, W, H, C.byref(s), L.TRACE_FAST_EXACT, full.device, None))
            ctx.record(e1)
            exact_ms = ctx.elapsed_ms(e0, e1) / reps
            exact_equal = bool((full.get_data().view(np.uint32) == ref_frame).all())
            if not exact_equal:
                raise SystemExit("LBVH_TRACE_FAST_EXACT frame differs from the LBVH_TRACE_REFERENCE frame")
            exact_mode = {"mode": "LBVH_TRACE_FAST_EXACT", "trace_ms": round(exact_ms, 4), "Mrays_s": round(W * H / (exact_ms * 1e-3) / 1e6, 2),
                          "equals_reference_mode_word_for_word": exact_equal,
                          "pixels_where_plain_fast_mode_differs": fast_differs,
                          "note": "the timed mode is LBVH_TRACE_FAST (same t everywhere; on exact t ties it keeps the lowest triangle "
                                  "index, the reference the triangle its visit order meets first)"}
            kernel_ms_of(ccam)

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

        # measured HBM copy rate of this box (float4 copy, 1 GiB) and the shader clock it holds under vector-ALU load
        nbytes = 1 << 30
        a = DataBuffer(ctx, nbytes // 4, np.uint32)
        b = DataBuffer(ctx, nbytes // 4, np.uint32)
        a.fill_u32(1)
        for _ in range(2):
            ctx.copy_probe(b.device, a.device, nbytes)
        ctx.record(e0)
        for _ in range(10):
            ctx.copy_probe(b.device, a.device, nbytes)
        ctx.record(e1)
        copy_gbs = 2.0 * nbytes * 10 / (ctx.elapsed_ms(e0, e1) * 1e-3) / 1e9
        a.dispose(); b.dispose()
        clock_mhz = ctx.clock_probe()

        # ---- the roofline object ----------------------------------------------------------------------
        own_bpr = own_bytes(st) if mode == L.TRACE_FAST else None
        bytes_per_ray = own_bpr if own_bpr is not None else ref_bytes(rs)
        hbm_achieved = bytes_per_ray * W * H / (trace_kernel_ms * 1e-3) / 1e9
        hbm = {"achieved": round(hbm_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 4),
               "bytes_per_ray": round(bytes_per_ray, 1),
               "bytes_per_ray_basis": "the kernel's own algorithm: 64 B per node line + 64 B per triangle line per 64-ray packet "
                                      "+ 16 B hit record per ray, from this run's visit counters" if own_bpr is not None else
                                      "reference visit order, 32P+24B+44L+48T+8 (SURVEY 8d)",
               "measured_copy_GBs": round(copy_gbs, 1), "frac_of_measured_copy": round(hbm_achieved / copy_gbs, 4)}
        live = (world == 1 and not args.no_live_counters and not cfg4 and mode == L.TRACE_FAST)
        traffic = issue = None
        if live:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import live_counters as LC
            child = [os.path.join(ROOT, "tools", "trace_only.py"), "--reps", "4", "--no-check"]
            traffic = LC.hbm_traffic(child, "trace_")
            issue = LC.issue_counters(child, "trace_")
        # HBM is the roofline of every stage.  `frac` is the kernel's OWN algorithmic bytes over its live duration
        # against 8 TB/s — small, because the walk prunes (the reference algorithm's bytes priced at this duration would be
        # several times the peak: `reference_equivalent_GBs`) and because the scene it walks is cache-resident.  The busiest unit
        # of the kernel is the vector pipe, carried beside it under its own name (`valu_issue`); DESIGN 13.3 for what binds a step.
        # roofline_version 4 (round 5): `bound` says what DESIGN 13.3 measured — a step is a ~90-instruction
        # dependent chain across the vector pipe (busy 0.9), the scalar unit and the memory system: "issue+latency" — while
        # achieved / peak / frac stay the HBM figures SURVEY 8(d) asks for (`roofline_of`); the counters that matter are also
        # flat scalars (traffic_bytes, valu_pipe_busy_frac, ...) beside the nested objects
        roofline = {"roofline_version": 4,
                    "kernel": kname, "kernel_ms": round(trace_kernel_ms, 4), "bound": "issue+latency", "roofline_of": "hbm",
                    "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hbm["frac"],
                    "algorithmic_bytes": round(bytes_per_ray * W * H), "bytes_per_ray": hbm["bytes_per_ray"],
                    "bytes_per_ray_basis": hbm["bytes_per_ray_basis"], "measured_copy_GBs": hbm["measured_copy_GBs"],
                    "frac_of_measured_copy": hbm["frac_of_measured_copy"], "clock_MHz": round(clock_mhz, 1)}
        if issue and issue.get("SQ_INSTS_VALU"):
            valu = issue["SQ_INSTS_VALU"]
            peak = SIMDS * clock_mhz * 1e6 / VALU_ISSUE_CYCLES / 1e9              # G wave-instructions per second
            achieved = valu / (trace_kernel_ms * 1e-3) / 1e9
            steps = float(st["pops"])
            simd_cycles = SIMDS * clock_mhz * 1e6 * trace_kernel_ms * 1e-3
            roofline["valu_issue"] = {
                "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "G wave-instr/s", "frac": round(achieved / peak, 4),
                "basis": "SQ_INSTS_VALU of a live rocprofv3 --pmc child pass of this run (the same frame, cost-ordered dispatch) / the "
                         "kernel's live HIP-event duration; peak = 1024 SIMDs x the measured shader clock / 2 cycles per wave64 vector "
                         "instruction (MI355X_MICROARCH.md: v_fma_f32 2 cycles on the SIMD-32 with other waves resident).  Only v_mul / "
                         "v_add / v_fma / v_mov reach that rate on this chip; the walk's v_cmp, min3 / max3, DPP-operand and v_readlane "
                         "instructions issue in 4.3 cycles (profiles/r3/b_instcost.txt): `pipe_busy_frac` is the share of the kernel's "
                         "SIMD cycles with a vector instruction executing (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles).  The busiest unit, not "
                         "the whole bound: 17 % fewer of these instructions (slab products on the matrix pipe) shorten the frame by "
                         "0 - 2 % (profiles/r4/d_matrix_pipe_products.txt)",
                "pipe_busy_frac": round(4.0 * issue.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles, 3),
                "cycles_per_valu": round(4.0 * issue.get("SQ_ACTIVE_INST_VALU", 0.0) / valu, 2),
                "valu_per_step": round(valu / steps, 1), "salu_per_step": round(issue.get("SQ_INSTS_SALU", 0.0) / steps, 1),
                "steps": int(steps),
                "lane_utilisation": round(issue.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(issue.get("SQ_ACTIVE_INST_VALU", 1.0), 1.0) / 64.0, 3),
                "salu_issue_frac": round(issue.get("SQ_INSTS_SALU", 0.0) / (256 * clock_mhz * 1e6 * trace_kernel_ms * 1e-3), 4),
                "waves_waiting_frac": round(issue.get("SQ_WAIT_ANY", 0.0) / max(issue.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)}
        else:
            roofline["valu_issue"] = None
        roofline["traffic"] = traffic
        # the same as flat scalars (a parser that keeps one level of the object keeps these)
        roofline["traffic_bytes"] = traffic["bytes"] if traffic else None
        roofline["traffic_fetch_bytes"] = traffic["fetch_bytes_x2"] if traffic else None
        roofline["traffic_write_bytes"] = traffic["write_bytes"] if traffic else None
        roofline["traffic_over_algorithmic"] = round(traffic["bytes"] / (bytes_per_ray * W * H), 3) if traffic else None
        vi = roofline["valu_issue"]
        for k_flat, k_src in (("valu_pipe_busy_frac", "pipe_busy_frac"), ("valu_issue_frac", "frac"), ("valu_per_step", "valu_per_step"),
                              ("salu_per_step", "salu_per_step"), ("steps_per_frame", "steps"), ("cycles_per_valu", "cycles_per_valu"),
                              ("salu_issue_frac", "salu_issue_frac"), ("lane_utilisation", "lane_utilisation")):
            roofline[k_flat] = vi[k_src] if vi else None
        # what the kernel replaces: the reference algorithm's per-ray walk, priced at this kernel's duration (a speed-up
        # figure, not a roofline fraction)
        roofline["reference_equivalent_GBs"] = round(ref_bytes(rs) * W * H / (trace_kernel_ms * 1e-3) / 1e9, 1)
        roofline["reference_bytes_per_ray"] = round(ref_bytes(rs), 1)
        roofline["reference_visits_per_ray"] = ref_counts

        sort_roofline = None
        if not args.no_sort_bench:
            sort_roofline = sort_microbench(ctx, args.sort_keys_log2, copy_gbs, live and args.sort_keys_log2 == 26)

        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline and not cfg4:
            cpu_baseline = cpu_leg(tris, cam)

        dynamic = None
        if world == 1 and not args.no_dynamic and not cfg4:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import dynamic_bench                               # cfg5: animate + rebuild + primary + 4 bounces per frame
            dynamic = dynamic_bench.run(ctx, frames=10, warmup=2, live=live)

        out = {
            "metric": "LBVH build Mtri/s + primary Mrays/s at 1080p on 1M-tri synthetic mesh",
            "value": round(W * H / (trace_ms_max * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "build_Mtri_s": round(n_tris / (build_ms_max * 1e-3) / 1e6, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall_ms, 4),
            # the split of a step (and with it `value` and build_Mtri_s) comes from an instrumented pass over the same K steps (two
            # event records per step) that runs BEFORE the timed region; `ms_per_step` is the wall clock of K steps with nothing else
            # on the stream
            "ms_per_step_instrumented": round(wall_instrumented_ms, 4),
            "build_ms": round(build_ms_max, 4), "trace_ms": round(trace_ms_max, 4),
            # `value` is the best of three cases (every timed frame is dispatched from the identical previous frame's per-tile
            # costs): the same share as a FIRST frame and under a camera turning 1 degree per frame, first-class beside it
            "value_cold": round(W * H / (extras["trace_cold_ms"] * 1e-3) / 1e6, 2) if extras else None,
            "value_moving": round(W * H / (extras["trace_moving_ms"] * 1e-3) / 1e6, 2) if extras else None,
            "value_static_scene": round(W * H / (extras["trace_static_scene_ms"] * 1e-3) / 1e6, 2) if extras else None,
            "trace_cold_ms": extras.get("trace_cold_ms"), "trace_moving_ms": extras.get("trace_moving_ms"),
            "trace_static_scene_ms": extras.get("trace_static_scene_ms"),
            "build_reference_stages_ms": round(ref_build_ms, 4),
            "build_reference_stages_Mtri_s": round(n_tris / (ref_build_ms * 1e-3) / 1e6, 2),
            "value_without_gather": round(W * H / (own_trace_ms_max * 1e-3) / 1e6, 2),
            "trace_without_gather_ms": round(own_trace_ms_max, 4),
            "frame_gather": frame_check,
            # N > 1: which way the shares reached rank 0 and how many ranks the collective library reported (flat: VERDICT r5 item 4)
            "frame_gather_transport": gather.mode if gather is not None else None,
            "rccl_ranks": (dist.get_world_size() if dist is not None and args.backend == "nccl" else None),
            # parity guards of THIS run, flat (the line is not printed unless both hold): container arrays after the timed steps ==
            # an independent staged rebuild, word for word; timed frame's t == LBVH_TRACE_REFERENCE's, bit for bit
            "timed_path_arrays_equal_staged_chain": timed_path_check["arrays_equal_staged_four_pass_chain"] if timed_path_check else None,
            "t_mismatch_vs_reference_mode": t_mismatch,
            "timed_path_check": timed_path_check,
            # methodology marker: 2 = rounds 5+: defaults --steps 100 --warmup 10; value / build_ms / trace_ms from an
            # instrumented pass BEFORE the timed region, ms_per_step from the timed region (rounds 1-4: 20 / 3, one pass);
            # 3 = round 6: + the two parity guards above
            "bench_version": 3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "scaling_note": "strong scaling of ONE 1080p frame: `value` = rays of the whole frame / the slowest rank's trace part — for N > 1 "
                            "that part ends with the whole frame in rank 0's buffer (`frame_gather`; `value_without_gather`: the traversal "
                            "alone).  The rebuild is replicated (every rank builds the whole BVH: north_star's 'BVH replicated, no collectives'), so "
                            f"ms_per_step cannot fall below build_ms ({build_ms_max:.3f} ms here) however many GPUs trace: at 8 GPUs the step "
                            "is at best ~1.4x faster than on one.  A share's trace time is the dependent chain of its heaviest 8x8 tile "
                            "(~0.4 us per step), not throughput: expect ~0.8 / 0.45 / 0.3 efficiency at 2 / 4 / 8 GPUs for `value` "
                            "(DESIGN.md sections 6 and 13.1, profiles/r4/n_shard_times_one_gpu.txt)",
            "dtype": "f32+u32", "data": "synthetic",
            "config": {"workload": ("cfg4: 16,000,000-triangle tiled bumpy torus (400x160 quads x 125 tiles, seed 2), sort "
                                    + (f"key-range sharded over {world} GPUs (RCCL digit-histogram all-reduce + one all-to-all), "
                                       if world > 1 else "on one GPU, ") + "1920x1080 primary rays; full LBVH rebuild + frame trace per step")
                       if cfg4 else
                                   "cfg2: 1,000,000-triangle tiled bumpy torus (seed 2), 1920x1080 primary rays, "
                                   "camera (0,0,250) fov 60; full LBVH rebuild + frame trace per step",
                       "triangles": n_tris, "rays": W * H, "trace_mode": args.mode,
                       "sharding": f"rays in interleaved groups of 8 tiles over {world} GPU(s) (one launch per GPU), BVH replicated, "
                                   "no collective in the traversal" + ("" if gather is None else
                                   "; every step ends with the whole frame in rank 0's buffer: " +
                                   ("shares traced straight into rank 0's memory (IPC-mapped, xGMI stores) + device-side completion flags"
                                    if gather.mode == "peer" else "packed shares + one torch.distributed gather + lbvh_frame_unpack")),
                       "hit_fraction": round(hit_fraction, 4)},
            "roofline": roofline,
            "roofline_sort_scatter": sort_roofline,
            "cpu_baseline": cpu_baseline,
            "trace_variants_ms": dict(extras, note="this rank's share of the frame, HIP events around the trace alone, max over ranks: "
                                                   "static = frame after frame without the rebuild; cold = dispatch history dropped "
                                                   "before every frame (what a first frame costs); moving = camera yawed 1 degree per frame")
                                 if extras else None,
            "second_camera": near,
            "exact_mode": exact_mode,
            "build_kernels_ms": prof_build,
            "cfg5_dynamic": dynamic,
        }
        if sharded_sort_check is not None:
            out["sharded_sort_matches_single_gpu"] = sharded_sort_check
        if weak is not None:
            out["weak_scaling_extra"] = weak
        if frame_4k is not None:
            out["frame_4k"] = frame_4k
            out["value_4k"] = frame_4k["Mrays_s"]
        if in_flight is not None:
            out["steps_in_flight"] = in_flight
    for e in events + [(ev_start,)]:
        for x in e:
            ctx.destroy_event(x)
    if gather is not None:
        gather.close()
    drawer.on_destroy()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


def sort_microbench(ctx, log2n, copy_gbs, live):
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
    traffic = None
    if live:
        import live_counters as LC
        traffic = LC.hbm_traffic([os.path.join(ROOT, "tools", "sort_bench.py"), str(log2n)], "sort_onesweep_kernel", pick="mean")
    return {"kernel": "sort scatter pass", "keys": n, "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "algorithmic_bytes": 16 * n,
            "kernel_ms": round(kernel_ms, 4), "measured_copy_GBs": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
            "sort_Gkeys_s": round(n / (sort_ms * 1e-3) / 1e9, 3), "sort_ms": round(sort_ms, 3),
            "kernels_ms": {name: round(v[1] / reps, 4) for name, v in prof_sum.items()}}


def cpu_leg(tris, cam):
    """CPU baseline (kind "port"): the oracle — a C restatement of the reference's C#/HLSL, the only runnable form of it
    here (no dotnet/mono/dxc) — on this box's host cores.  Two figures, as north_star asks: scalar (1 thread) and OpenMP
    (all cores: parallel-for over triangles / internal nodes / 8x8 ray tiles, parallel LSD radix sort, two-level
    DistributeKeys scan, flag hand-off refit).  Bounded sample: warm full 1 M-triangle rebuilds, then the 1080p frame
    subsampled on a pixel grid chosen from a pilot run so each traversal leg takes about 6 s."""
    import oracle as O
    cap = ((len(tris) + 1023) // 1024) * 1024
    b = O.Built(tris, capacity=cap, threads=min(O.num_threads(), 16))          # allocates and touches every array once
    # threads actually worth using: the box may show more hardware threads than this process is allowed to run on
    # (cgroup quota, affinity), and barrier-heavy OpenMP phases collapse when oversubscribed — pick the count that
    # builds fastest
    allowed = min(O.num_threads(), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else O.num_threads())
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            allowed = max(1, min(allowed, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    candidates = sorted({c for c in (4, 8, 16, 32, 64, 128, 256, allowed) if c <= allowed} | {allowed})
    threads = min(candidates, key=lambda c: min(b.rebuild(c) for _ in range(2)))

    def build_rate(th, budget_s):
        t0, secs = time.perf_counter(), []
        while len(secs) < 20 and (time.perf_counter() - t0 < budget_s or len(secs) < 2):
            secs.append(b.rebuild(th))
        return len(secs), min(secs)

    def trace_rate(th, budget_s):
        p0 = time.perf_counter()
        pilot, _ = O.trace_primary(b, cam, step=(32, 32), threads=th)
        rate = pilot.size / max(time.perf_counter() - p0, 1e-6)
        step = 1
        for s_ in (1, 2, 4, 8, 16, 32):
            step = s_
            if (W // s_) * (H // s_) / rate <= budget_s:
                break
        t1 = time.perf_counter()
        hits, _ = O.trace_primary(b, cam, step=(step, step), threads=th)
        dt = time.perf_counter() - t1
        return hits.size / dt / 1e6, step, hits.size, dt

    nb1, s1 = build_rate(1, 3.0)
    nbt, st = build_rate(threads, 3.0)
    r1, step1, n1, dt1 = trace_rate(1, 6.0)
    # the trace leg is embarrassingly parallel (8x8 ray tiles handed out dynamically, no barriers): every core the process may
    # use, not the count that suits the barrier-heavy build
    rt, stept, nt, dtt = trace_rate(allowed, 6.0)
    return {"value": round(rt, 4), "unit": "Mrays/s", "build_Mtri_s": round(len(tris) / st / 1e6, 4), "cores": allowed, "kind": "port",
            "nproc": os.cpu_count(), "cores_allowed": allowed, "trace_cores": allowed, "build_cores": threads,
            "scalar": {"value": round(r1, 4), "unit": "Mrays/s", "build_Mtri_s": round(len(tris) / s1 / 1e6, 4), "cores": 1},
            "sample": f"the box shows {os.cpu_count()} hardware threads, this process may use {allowed}; OpenMP: best of {nbt} warm 1M-triangle "
                      f"rebuilds on {threads} threads (the count that builds fastest: {st:.4f} s) + the 1080p frame on {allowed} threads sampled "
                      f"every {stept} pixel(s) in x and y ({nt} rays, {dtt:.2f} s); scalar: best of {nb1} rebuilds ({s1:.3f} s) + every {step1} "
                      f"pixel(s) ({n1} rays, {dt1:.2f} s); reference visit order (no pruning), 8x8 ray tiles handed out dynamically"}


if __name__ == "__main__":
    main()
