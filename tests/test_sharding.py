"""The N>1 path on CPU: world_size-2 gloo run of bench.py's ray sharding (interleaved tile groups per rank,
BVH replicated, no data-path collective) with the oracle standing in for the GPU kernels.  Checks that
the shards partition the frame, that per-rank pieces stitch to the unsharded frame, and that the
max-over-ranks timing reduction works over gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import shard_tiles  # noqa: E402


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("size", [(16, 8), (250, 131), (1920, 1080), (1921, 1083)])
def test_shards_partition_the_frame(world, size):
    width, height = size
    covered = np.zeros((height, width), dtype=np.int32)
    counts = []
    for r in range(world):
        tiles = shard_tiles(r, world, width, height)
        counts.append(len(tiles))
        for x0, y0, x1, y1 in tiles:
            assert 0 <= x0 < x1 <= width and 0 <= y0 < y1 <= height and x0 % 8 == 0 and y0 % 8 == 0
            covered[y0:y1, x0:x1] += 1
    assert (covered == 1).all()
    assert max(counts) - min(counts) <= 8             # balanced to one group of tiles


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    import oracle as O
    from unitysimpleraytracing_amd import layouts as L
    from unitysimpleraytracing_amd import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 96, 80
    tris = scenes.random_triangles(2000, seed=7)          # every rank builds the same tree (replicated)
    b = O.Built(tris, capacity=2048)
    cam = scenes.camera(W, H, (0.0, 0.0, 300.0))
    mine = np.zeros((H, W), dtype=np.float32)
    owned = np.zeros((H, W), dtype=np.float32)
    for x0, y0, x1, y1 in shard_tiles(rank, world, W, H):
        hits, _ = O.trace_primary(b, cam, rect=(x0, y0, x1, y1))
        mine[y0:y1, x0:x1] = hits["t"]
        owned[y0:y1, x0:x1] = 1.0
    t_mine = torch.from_numpy(mine * owned)
    t_owned = torch.from_numpy(owned)
    dist.all_reduce(t_mine)                                # test-side gather only; the data path has none
    dist.all_reduce(t_owned)
    ms = torch.tensor([float(10 + rank)], dtype=torch.float64)
    dist.all_reduce(ms, op=dist.ReduceOp.MAX)              # bench.py's max-over-ranks timing
    if rank == 0:
        full, _ = O.trace_primary(b, cam)
        ok = bool((t_owned.numpy() == 1.0).all() and (t_mine.numpy() == full["t"]).all() and ms.item() == 10 + world - 1)
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


# ---- the gather: every rank's records into ONE frame on rank 0 -----------------------------------

@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("size", [(8, 8), (250, 131), (1920, 1080), (1921, 1083)])
def test_packed_share_layout_mirrors_the_library(world, size):
    """frame_gather's numpy mirror of the packed layout agrees with the library's pure size function, shares partition the
    frame's tiles, and pack -> unpack is the identity on every pixel (odd sizes: partial tiles at the right / bottom edge)."""
    from unitysimpleraytracing_amd import _native as N
    from unitysimpleraytracing_amd import frame_gather as G
    width, height = size
    n_tiles = ((width + 7) // 8) * ((height + 7) // 8)
    seen = []
    for r in range(world):
        items = G.share_items(r, world, width, height)
        assert len(items) * 64 == G.shard_records(width, height, r, world) == N.lib.lbvh_shard_records(width, height, r, world)
        seen += [int(t) for t in items if t >= 0]
    assert sorted(seen) == list(range(n_tiles))
    assert N.lib.lbvh_shard_records(width, height, world, world) == 0 and N.lib.lbvh_shard_records(0, height, 0, world) == 0
    if width * height <= 300 * 200:
        rng = np.random.default_rng(5)
        frame = rng.integers(1, 1 << 30, size=(height, width)).astype(np.uint32)
        shares = [G.pack_share(frame, r, world) for r in range(world)]
        assert (G.unpack_shares(shares, world, width, height) == frame).all()


def _gather_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    import oracle as O
    from unitysimpleraytracing_amd import frame_gather as G
    from unitysimpleraytracing_amd import layouts as L
    from unitysimpleraytracing_amd import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 101, 77                                         # partial tiles on both edges, a ragged last group
    tris = scenes.random_triangles(2000, seed=9)
    b = O.Built(tris, capacity=2048)
    cam = scenes.camera(W, H, (0.0, 0.0, 300.0))
    mine = np.zeros((H, W), dtype=L.HIT)
    mine.view(np.uint32)[:] = 0x7FC00000                   # what this rank does not own is poison
    for x0, y0, x1, y1 in shard_tiles(rank, world, W, H):
        hits, _ = O.trace_primary(b, cam, rect=(x0, y0, x1, y1))
        mine[y0:y1, x0:x1] = hits
    packed = G.pack_share(mine, rank, world)
    stride = G.shard_records(W, H, 0, world)               # the largest share: equal blocks for the collective
    block = np.zeros(stride, dtype=L.HIT)
    block[:len(packed)] = packed
    t = torch.from_numpy(block.view(np.uint32).reshape(-1).copy().view(np.int32))
    parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, parts, dst=0)                           # what bench.py's packed transport does over RCCL
    if rank == 0:
        shares = [p.numpy().view(np.uint32).view(L.HIT) for p in parts]
        frame = np.zeros((H, W), dtype=L.HIT)
        frame.view(np.uint32)[:] = 0x7FC00000
        G.unpack_shares(shares, world, W, H, frame=frame)
        full, _ = O.trace_primary(b, cam)
        ok = bool((frame.view(np.uint32) == full.view(np.uint32)).all())
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_gather_one_whole_frame_over_gloo(tmp_path, world):
    """The N > 1 step ends with ONE frame on rank 0: every rank's share, packed, through one gather, unpacked at its
    pixels == the frame traced whole, every word of every record (oracle as the tracer; the same layout functions the
    GPU kernels are tested against in test_gpu_parity.py)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gather.txt")
    mp.spawn(_gather_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_bench_starts_its_own_ranks_for_gpus_n():
    """`python bench.py --gpus N` from a plain launch starts N ranks itself (children with torchrun's
    environment, created before the parent touches HIP) and ends with their worst return code.  No GPU here: every rank
    must fail loudly in lbvh_create ("no HIP device") — never fall back to a CPU path — and the parent must report it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible: the ranks would run the real benchmark")
    except ImportError:
        pass
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr and r.stdout.strip() == ""           # no JSON line from a run that did not run
    assert r.stderr.count("lbvh_create") >= 1


def _bench(args, extra_env=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_rccl_launch_fails_cleanly_without_gpus():
    """the launch shape the driver uses on an 8-GPU node — `--gpus N`, backend nccl (= RCCL), one rank per
    GPU, `device_id=` given to init_process_group — has only ever run with gloo here.  Without GPUs it must end at once with
    a status and one sentence per rank that names the missing device (counted BEFORE anything initialises HIP), not with a
    traceback from inside torch or a hang in a rendezvous; and never with a JSON line."""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            pytest.skip("a GPU is visible: the ranks would run the real benchmark")
    except ImportError:
        pytest.skip("no torch")
    import time
    t0 = time.time()
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])                  # self-launched ranks, file:// rendezvous
    assert r.returncode != 0 and r.stdout.strip() == "" and "Traceback" not in r.stderr
    assert r.stderr.count("needs HIP device") == 2 and "rank 1/2: needs HIP device 1" in r.stderr
    # the driver's shape: torchrun's environment, one process = one rank
    env = {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29544"}
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert r.returncode != 0 and "rank 1/2: needs HIP device 1" in r.stderr and "Traceback" not in r.stderr
    # arguments that cannot mean anything
    r = _bench(["--gpus", "4", "--steps", "1"], dict(env, RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "Traceback" not in r.stderr
    r = _bench(["--gpus", "2", "--backend", "nccl", "--device", "0", "--steps", "1"], env)
    assert r.returncode != 0 and ("one GPU per rank" in r.stderr or "needs HIP device" in r.stderr) and "Traceback" not in r.stderr
    assert time.time() - t0 < 120


def test_bench_launch_has_an_overall_deadline():
    """if ALL ranks hang (rendezvous, RCCL's set-up, a collective) nobody fails and the ten seconds' grace never
    starts — the parent of `bench.py --gpus N` must end them itself when --launch-deadline passes: SIGTERM, SIGKILL five seconds
    later for a rank that ignores it (rank 1 of the hook does), exit status 124, no JSON line, the hung ranks named."""
    import time
    t0 = time.time()
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-deadline", "3"], {"LBVH_BENCH_TEST_HANG": "1"}, timeout=120)
    took = time.time() - t0
    assert r.returncode == 124 and r.stdout.strip() == "" and "Traceback" not in r.stderr
    assert "did not finish within 3 s" in r.stderr and "[0, 1]" in r.stderr
    assert "HSA_ENABLE_IPC_MODE_LEGACY=" in r.stderr
    assert 3.0 <= took < 40.0, took
    # nothing of the launch is left behind
    import subprocess
    ps = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    left = [l for l in ps.splitlines() if "bench.py --gpus 2 --steps 1 --warmup 0 --launch-deadline 3" in l]
    assert not left, left


def test_centre_out_tile_order_is_a_permutation():
    """Frames without dispatch history take rows and columns (a share: its groups of 8 tiles) from the middle outwards
    (csrc/lbvh_trace.hip trace_packet_kernel, centre_out): the arithmetic, restated, is a bijection for every size — every tile
    of a whole frame and every work item of a share is traced exactly once, the slots past the last tile stay where they are."""
    def centre_out(k, n):
        mid = n // 2
        if n & 1:
            return mid + (k + 1) // 2 if k & 1 else mid - k // 2
        return mid + (k - 1) // 2 if k & 1 else mid - 1 - k // 2

    for n in list(range(1, 70)) + [135, 240, 1023, 1024]:
        assert sorted(centre_out(k, n) for k in range(n)) == list(range(n))
        assert centre_out(0, n) in (n // 2, n // 2 - 1 + (n & 1))           # starts in the middle
    for tiles_x, tiles_y in ((240, 135), (1, 1), (7, 3), (32, 17)):
        n_tiles = tiles_x * tiles_y
        n_work = (n_tiles + 7) // 8 * 8                                      # whole frame: work item = tile, the last group's tail beyond
        out = []
        for w in range(n_work):
            if w < n_tiles:
                ty, tx = divmod(w, tiles_x)
                out.append(centre_out(ty, tiles_y) * tiles_x + centre_out(tx, tiles_x))
            else:
                out.append(w)
        assert sorted(out) == list(range(n_work))
    for groups in (1, 2, 5, 506, 1013):                                      # a share: groups of 8 work items
        out = [centre_out(w // 8, groups) * 8 + w % 8 for w in range(groups * 8)]
        assert sorted(out) == list(range(groups * 8))
