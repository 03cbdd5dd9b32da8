"""cfg4 (BASELINE configs[3]): key-range sharded radix sort, one process per GPU.

The reference sorts on one device (Assets/_Scripts/ComputeBufferSorter.cs:100-126); this is the
multi-GPU form of the same stable (key, value) sort.  MI355X-first shape: xGMI is point-to-point, so
the data crosses it ONCE (one all-to-all of key ranges, all 7 links busy) instead of once per LSD
pass; the only RCCL all-reduces are the [W-1][256] MSD digit histograms that pick the splitters.

    1. every rank sorts its own block                       lbvh_sort_pairs      (local, HBM-bound)
    2. ranks agree on the W-1 splitter keys = the order     lbvh_key_histogram   (local)
       statistics at q*N/W: four MSD rounds, each one        all_reduce(SUM)      (RCCL, <= 28 KB)
       all-reduce of the 8-bit digit histograms of the keys
       that share the prefix found so far
    3. send offsets of the splitters in the sorted block    lbvh_lower_bound     (local)
    4. one exchange: rank q receives every key in            all_to_all_single    (RCCL over xGMI)
       [splitter_q, splitter_q+1) from every rank, in rank order
    5. every rank sorts what it received                    lbvh_sort_pairs      (local)

The concatenation of the ranks' results is bit-identical to lbvh_sort_pairs over the whole array:
equal keys always land on one rank, arrive ordered by (source rank, position in the source's stably
sorted block) = original global order, and step 5 is stable.

Host logic only: every computation over keys is a C-ABI call (HipKeyOps).  `ops` is injectable so
the world_size-2 gloo tests on CPU can drive the same logic with the oracle standing in for the
kernels; the product constructor takes a Context and has no fallback.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N

DIGIT_BITS = 8
LEVEL_SHIFTS = (24, 16, 8, 0)          # MSD first


class HipKeyOps:
    """The local kernels, on CUDA tensors (device pointers handed to the C ABI).  The context must
    have been created on torch's current stream (Context(device, stream=torch.cuda.current_stream().cuda_stream))
    so that kernels and RCCL collectives are ordered by torch's stream semantics."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device_id)

    def sort_pairs(self, keys, vals):
        assert keys.dtype == torch.int32 and vals.dtype == torch.int32 and keys.is_cuda and keys.is_contiguous()
        N.check(self.ctx.handle, N.lib.lbvh_sort_pairs(self.ctx.handle, C.c_void_p(keys.data_ptr()),
                                                       C.c_void_p(vals.data_ptr()), keys.numel()))

    def key_histogram(self, keys, prefixes, prefix_shift, shift):
        n_p = 1 if prefixes is None else len(prefixes)
        hist = torch.empty(n_p * 256, dtype=torch.int32, device=self.device)
        arr = None if prefixes is None else (C.c_uint32 * n_p)(*[int(p) for p in prefixes])
        N.check(self.ctx.handle, N.lib.lbvh_key_histogram(self.ctx.handle, C.c_void_p(keys.data_ptr()), keys.numel(),
                                                          arr, n_p, prefix_shift, shift, C.c_void_p(hist.data_ptr())))
        return hist.view(n_p, 256)

    def lower_bound(self, sorted_keys, probes):
        out = torch.empty(len(probes), dtype=torch.int32, device=self.device)
        arr = (C.c_uint32 * len(probes))(*[int(p) for p in probes])
        N.check(self.ctx.handle, N.lib.lbvh_lower_bound(self.ctx.handle, C.c_void_p(sorted_keys.data_ptr()),
                                                        sorted_keys.numel(), arr, len(probes), C.c_void_p(out.data_ptr())))
        return out

    def empty(self, n):
        return torch.empty(n, dtype=torch.int32, device=self.device)


class _Comm:
    """torch.distributed with one test hook: under gloo (ranks sharing a GPU, or CPU tests) device
    tensors are staged through the host, since gloo moves host memory."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) == "gloo"

    def all_reduce_sum(self, t):
        if self.staged and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def all_gather_host_ints(self, values):
        """Every rank's small list of Python ints -> [world][len] list (a host sync, like the split sizes)."""
        t = torch.tensor(values, dtype=torch.int64)
        if not self.staged:
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return [o.tolist() for o in out]

    def all_to_all(self, send, send_counts, recv, recv_counts):
        if self.staged:
            if send.is_cuda:
                hs, hr = send.cpu(), torch.empty(recv.numel(), dtype=recv.dtype)
            else:
                hs, hr = send, recv
            # gloo has no all_to_all_single with uneven splits on every build: W point-to-point rounds
            s_off = np.concatenate([[0], np.cumsum(send_counts)])
            r_off = np.concatenate([[0], np.cumsum(recv_counts)])
            reqs = []
            for peer in range(self.world):
                if peer == self.rank:
                    hr[r_off[peer]: r_off[peer + 1]] = hs[s_off[peer]: s_off[peer + 1]]
                    continue
                if send_counts[peer]:
                    reqs.append(dist.isend(hs[s_off[peer]: s_off[peer + 1]].contiguous(), peer, group=self.group))
            for peer in range(self.world):
                if peer != self.rank and recv_counts[peer]:
                    buf = torch.empty(recv_counts[peer], dtype=recv.dtype)
                    dist.recv(buf, peer, group=self.group)
                    hr[r_off[peer]: r_off[peer + 1]] = buf
            for r in reqs:
                r.wait()
            if recv.is_cuda:
                recv.copy_(hr)
        else:
            dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                                   group=self.group)

    def all_gather_ragged(self, local, counts):
        """Concatenation over ranks of `local` (counts[r] elements on rank r)."""
        width = max(max(counts), 1)
        if self.staged:
            h = local.cpu() if local.is_cuda else local
            padded = torch.zeros(width, dtype=h.dtype)
            padded[: h.numel()] = h
            slabs = [torch.empty(width, dtype=h.dtype) for _ in range(self.world)]
            dist.all_gather(slabs, padded, group=self.group)
            out = torch.cat([slabs[r][: counts[r]] for r in range(self.world)])
            return out.to(local.device) if local.is_cuda else out
        padded = torch.empty(width, dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local
        slab = torch.empty(width * self.world, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(slab, padded, group=self.group)
        return torch.cat([slab[r * width: r * width + counts[r]] for r in range(self.world)])


def _u32(x):
    return int(x) & 0xFFFFFFFF


class ShardedSorter:
    """ComputeBufferSorter for keys block-partitioned over the ranks of `group`.

    sort(keys, vals): keys/vals are this rank's block (int32 tensors holding the u32 bit patterns;
    vals are normally global triangle indices).  Returns (keys, vals, counts): this rank's slice of
    the globally sorted sequence (every key in [splitter_rank, splitter_rank+1)) and every rank's
    slice length.  gather() concatenates the slices on every rank (replicated tree build)."""

    def __init__(self, ctx=None, group=None, ops=None, always_exchange=False):
        self.always_exchange = always_exchange      # test hook: run the collectives even with one rank
        if ops is None:
            if ctx is None:
                raise ValueError("ShardedSorter needs a Context (HIP kernels) — there is no CPU path")
            ops = HipKeyOps(ctx)
        self.ops = ops
        self.comm = _Comm(group)
        self.splitters = None

    # -- step 2 ------------------------------------------------------------------------------------
    def find_splitters(self, keys, n_local_counts):
        """W-1 splitter keys: splitter_q = the key at global sorted position q*N/W (an order statistic,
        found MSD digit by digit from all-reduced histograms).  Identical on every rank."""
        W = self.comm.world
        if W == 1:
            return []
        total = int(sum(n_local_counts))
        targets = [(q * total) // W for q in range(1, W)]           # 0-based global positions
        prefixes = [0] * (W - 1)
        remaining = list(targets)
        for level, shift in enumerate(LEVEL_SHIFTS):
            if level == 0:
                hist = self.ops.key_histogram(keys, None, 32, shift)
            else:
                hist = self.ops.key_histogram(keys, prefixes, shift + DIGIT_BITS, shift)
            hist = hist.to(torch.int64) & 0xFFFFFFFF                 # u32 counts; the sum needs 64 bits
            hist = self.comm.all_reduce_sum(hist)                    # RCCL: the global digit histogram
            h = hist.cpu().numpy()                                   # [n_prefixes][256]
            for q in range(W - 1):
                row = h[0] if level == 0 else h[q]
                cum = np.cumsum(row)
                d = int(np.searchsorted(cum, remaining[q], side="right"))
                d = min(d, 255)
                remaining[q] -= int(cum[d - 1]) if d else 0
                prefixes[q] = (prefixes[q] << DIGIT_BITS) | d
        return [_u32(p) for p in prefixes]

    def sort(self, keys, vals):
        ops, comm = self.ops, self.comm
        W = comm.world
        n_local = keys.numel()
        ops.sort_pairs(keys, vals)                                                     # 1
        if W == 1 and not self.always_exchange:
            self.splitters = []
            return keys, vals, [n_local]
        counts = [c[0] for c in comm.all_gather_host_ints([n_local])]
        if sum(counts) == 0:
            return keys, vals, counts
        self.splitters = self.find_splitters(keys, counts)                             # 2
        pos = ops.lower_bound(keys, self.splitters).cpu().tolist()                      # 3
        edges = [0] + [int(p) & 0xFFFFFFFF for p in pos] + [n_local]
        send_counts = [edges[i + 1] - edges[i] for i in range(W)]
        assert min(send_counts) >= 0                                                   # splitters are non-decreasing
        table = comm.all_gather_host_ints(send_counts)                                 # [src][dst]
        recv_counts = [table[src][comm.rank] for src in range(W)]
        out_keys, out_vals = ops.empty(sum(recv_counts)), ops.empty(sum(recv_counts))
        comm.all_to_all(keys, send_counts, out_keys, recv_counts)                      # 4
        comm.all_to_all(vals, send_counts, out_vals, recv_counts)
        ops.sort_pairs(out_keys, out_vals)                                             # 5 (stable)
        return out_keys, out_vals, [sum(table[src][dst] for src in range(W)) for dst in range(W)]

    def gather(self, keys, vals, counts):
        if self.comm.world == 1:
            return keys, vals
        return self.comm.all_gather_ragged(keys, counts), self.comm.all_gather_ragged(vals, counts)


def block_of(rank, world, capacity):
    """[lo, hi) of rank's block of a capacity-long array (blocks of ceil(capacity / world))."""
    m = (capacity + world - 1) // world
    return min(rank * m, capacity), min((rank + 1) * m, capacity)


class DeviceArray:
    """A device pointer as a zero-copy torch tensor (__cuda_array_interface__): lets the sharded sort
    read and write DataBuffer storage."""

    def __init__(self, ptr, count, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}

    def tensor(self, device):
        return torch.as_tensor(self, device=device)


def sort_container(sorter, container):
    """ComputeBufferSorter.Sort() for a MeshBufferContainer whose keys / triangle indices were generated
    on every rank (Morton is replicated): each rank sorts its block through the sharded sorter and the
    gathered result is written back into the container's buffers, so DistributeKeys / ConstructTree /
    ConstructBVH run unchanged (replicas).  Returns the per-rank slice lengths."""
    comm = sorter.comm
    dev = sorter.ops.device
    cap = container.capacity
    keys = DeviceArray(container.keys.device.value, cap).tensor(dev)
    idx = DeviceArray(container.triangle_index.device.value, cap).tensor(dev)
    lo, hi = block_of(comm.rank, comm.world, cap)
    k, v, counts = sorter.sort(keys[lo:hi].clone(), idx[lo:hi].clone())
    gk, gv = sorter.gather(k, v, counts)
    keys.copy_(gk)
    idx.copy_(gv)
    return counts
