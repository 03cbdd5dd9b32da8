"""cfg4 (BASELINE configs[3]): key-range sharded radix sort, one process per GPU.

The reference sorts on one device (Assets/_Scripts/ComputeBufferSorter.cs:100-126); this is the
multi-GPU form of the same stable (key, value) sort.  MI355X-first shape: xGMI is point-to-point, so
the data crosses it ONCE (one all-to-all of key ranges, all 7 links busy) instead of once per LSD
pass; the only RCCL all-reduces are the [W-1][256] MSD digit histograms that pick the splitters.

    1. every rank sorts its own block                       lbvh_sort_pairs      (local, HBM-bound)
    2. ranks agree on the W-1 splitter keys = the order     lbvh_key_histogram_device (local)
       statistics at q*N/W: four MSD rounds, each one        all_reduce(SUM)      (RCCL, <= 28 KB)
       all-reduce of the 8-bit digit histograms of the keys
       that share the prefix found so far; the digit is
       picked on the device (torch ops on the small table)
    3. send offsets of the splitters in the sorted block    lbvh_lower_bound_device (local)
    4. one exchange: rank q receives every (key, value) in  all_to_all_single    (RCCL over xGMI, 64-bit words)
       [splitter_q, splitter_q+1) from every rank, in rank order
   One host synchronisation per sort (the [W][W] send-count table that all_to_all_single takes as host integers).
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
        """prefixes: None (every key) or an int64 DEVICE tensor of u32 values — they never visit the host"""
        n_p = 1 if prefixes is None else prefixes.numel()
        hist = torch.empty(n_p * 256, dtype=torch.int32, device=self.device)
        if prefixes is None:
            N.check(self.ctx.handle, N.lib.lbvh_key_histogram(self.ctx.handle, C.c_void_p(keys.data_ptr()), keys.numel(),
                                                              None, 1, prefix_shift, shift, C.c_void_p(hist.data_ptr())))
        else:
            p32 = prefixes.to(torch.int32)             # low 32 bits = the u32 pattern
            N.check(self.ctx.handle, N.lib.lbvh_key_histogram_device(self.ctx.handle, C.c_void_p(keys.data_ptr()), keys.numel(),
                                                                     C.c_void_p(p32.data_ptr()), n_p, prefix_shift, shift,
                                                                     C.c_void_p(hist.data_ptr())))
        return hist.view(n_p, 256)

    def lower_bound(self, sorted_keys, probes):
        """probes: an int64 DEVICE tensor of u32 values"""
        out = torch.empty(probes.numel(), dtype=torch.int32, device=self.device)
        p32 = probes.to(torch.int32)
        N.check(self.ctx.handle, N.lib.lbvh_lower_bound_device(self.ctx.handle, C.c_void_p(sorted_keys.data_ptr()), sorted_keys.numel(),
                                                               C.c_void_p(p32.data_ptr()), probes.numel(), C.c_void_p(out.data_ptr())))
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

    def all_gather_small(self, t):
        """Every rank's small 1-D int64 tensor -> [world][len] tensor on the same device (no host visit under RCCL)."""
        if self.staged and t.is_cuda:
            h = t.cpu()
            out = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(out, h, group=self.group)
            return torch.stack(out).to(t.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return torch.stack(out)

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
    def find_splitters(self, keys, total):
        """W-1 splitter keys: splitter_q = the key at global sorted position q*N/W (an order statistic, found MSD digit
        by digit from all-reduced histograms).  Identical on every rank.  Everything stays on the device: per round one
        histogram kernel, one all-reduce of the [W-1][256] table, and the digit selection as a handful of torch
        operations on that table (cumulative sum, search, gather) — no host round trip (round 2 read the table back to
        the host in every one of the four rounds).  Returns an int64 tensor of u32 values on the keys' device."""
        W = self.comm.world
        dev = keys.device
        if W == 1:
            return torch.empty(0, dtype=torch.int64, device=dev)
        remaining = torch.tensor([(q * total) // W for q in range(1, W)], dtype=torch.int64, device=dev)   # 0-based global positions
        prefixes = torch.zeros(W - 1, dtype=torch.int64, device=dev)
        for level, shift in enumerate(LEVEL_SHIFTS):
            if level == 0:
                hist = self.ops.key_histogram(keys, None, 32, shift)
            else:
                hist = self.ops.key_histogram(keys, prefixes, shift + DIGIT_BITS, shift)
            hist = hist.to(torch.int64) & 0xFFFFFFFF                 # u32 counts; the sum needs 64 bits
            hist = self.comm.all_reduce_sum(hist)                    # RCCL: the global digit histogram
            rows = hist.expand(W - 1, 256) if level == 0 else hist   # [W-1][256]
            cum = torch.cumsum(rows, dim=1)
            d = torch.searchsorted(cum.contiguous(), remaining.unsqueeze(1), right=True).squeeze(1).clamp(max=255)
            below = torch.where(d > 0, cum.gather(1, (d - 1).clamp(min=0).unsqueeze(1)).squeeze(1), torch.zeros_like(remaining))
            remaining = remaining - below
            prefixes = (prefixes << DIGIT_BITS) | d
        return prefixes

    def sort(self, keys, vals, counts=None):
        """counts: every rank's block length when the caller knows them (a fixed partition such as block_of): saves the
        one host round trip that gathers them.  Host synchronisations per sort: ONE (the [W][W] table of send counts,
        which all_to_all_single needs as host integers) + that optional one — round 2 had seven."""
        ops, comm = self.ops, self.comm
        W = comm.world
        n_local = keys.numel()
        ops.sort_pairs(keys, vals)                                                     # 1
        if W == 1 and not self.always_exchange:
            self.splitters = []
            return keys, vals, [n_local]
        if counts is None:
            counts = [c[0] for c in comm.all_gather_host_ints([n_local])]
        assert counts[comm.rank] == n_local
        if sum(counts) == 0:
            return keys, vals, list(counts)
        dev = keys.device
        splitters = self.find_splitters(keys, int(sum(counts)))                        # 2 (device)
        pos = ops.lower_bound(keys, splitters).to(torch.int64) & 0xFFFFFFFF if W > 1 else torch.empty(0, dtype=torch.int64, device=dev)   # 3
        edges = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), pos, torch.full((1,), n_local, dtype=torch.int64, device=dev)])
        send_dev = edges[1:] - edges[:-1]
        packed = torch.cat([send_dev, splitters])                                      # one small gather carries both
        table_dev = comm.all_gather_small(packed)                                      # [src][dst | splitters]
        table_h = table_dev.cpu()                                                      # the ONE host synchronisation
        table = table_h[:, :W].tolist()
        self.splitters = [_u32(x) for x in table_h[comm.rank, W:].tolist()]
        send_counts = table[comm.rank]
        assert min(send_counts) >= 0 and sum(send_counts) == n_local                   # splitters are non-decreasing
        recv_counts = [table[src][comm.rank] for src in range(W)]
        n_recv = sum(recv_counts)
        # 4: (key, value) pairs cross xGMI as ONE all-to-all of 64-bit words
        pairs = torch.stack([keys, vals], dim=1).contiguous().view(torch.int64).view(-1)
        out_pairs = torch.empty(n_recv, dtype=torch.int64, device=dev)
        comm.all_to_all(pairs, send_counts, out_pairs, recv_counts)
        kv = out_pairs.view(torch.int32).view(-1, 2)
        out_keys, out_vals = kv[:, 0].contiguous(), kv[:, 1].contiguous()
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
    blocks = [block_of(r, comm.world, cap) for r in range(comm.world)]
    k, v, counts = sorter.sort(keys[lo:hi].clone(), idx[lo:hi].clone(), counts=[b[1] - b[0] for b in blocks])
    gk, gv = sorter.gather(k, v, counts)
    keys.copy_(gk)
    idx.copy_(gv)
    return counts
