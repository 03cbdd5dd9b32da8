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
