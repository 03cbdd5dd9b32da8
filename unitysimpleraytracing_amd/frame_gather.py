"""One frame from N GPUs, one process per GPU (bench.py's launch shape; BASELINE configs[2], SURVEY 8e: "hit records
gathered to GPU 0").

The reference renders ONE image per Update() (Assets/_Scripts/RaytracingMeshDrawer.cs:76-89).  With the rays sharded over
the ranks (lbvh_trace_primary_shard, BVH replicated, no collective in the traversal) a frame is whole only when every
rank's hit records sit in one full-frame buffer on the rank that shades / displays it (rank 0).  Two transports, both
behind the C ABI's "one frame from N GPUs" calls (include/lbvh.h):

  peer     rank 0 exports its frame buffer and a flag array (lbvh_ipc_export), every other rank maps them
           (lbvh_ipc_import after lbvh_peer_enable) and traces its share STRAIGHT INTO rank 0's memory: the trace kernel's
           stores travel over xGMI while the tiles finish, nothing is copied afterwards.  Completion: rank r enqueues
           lbvh_frame_signal(flags[r - 1] := frame number) behind its trace, rank 0 enqueues lbvh_frame_wait — a bounded
           device-side wait — behind its own share; the other way round, rank 0 signals "frame f has been read" in front
           of its next share and the others wait for that before they store into the buffer again.  No host round trip and
           no collective per frame; the ranks have to enqueue their frames within the wait's bound (20 s of wall clock) of
           each other.  The flag words are uncached device memory (lbvh_flags_alloc): a kernel that polls them while another
           GPU stores into them must not be served from its own L2.  UNVERIFIED ON MORE THAN ONE PHYSICAL GPU
           until a multi-GPU run exists (every test and stand-in shares one GPU); `auto` keeps the self-test and the fallback.
  packed   every rank traces its share into a contiguous block (lbvh_trace_primary_shard_packed), the blocks reach rank 0
           by one torch.distributed gather (RCCL send / recv over xGMI), rank 0 puts them at their pixels
           (lbvh_frame_unpack).  The fallback when the GPUs cannot map each other's memory (or IPC is unavailable).

`auto` tries `peer` with a self-test (every rank writes a pattern into rank 0's buffer, signals, rank 0 checks it) and
falls back to `packed` — on ALL ranks together — if any rank fails.  Host logic only: every byte of the frame is moved
or produced by a C-ABI call or a torch.distributed collective; nothing here computes on hit records.

pack_share / unpack_shares are numpy mirrors of the packed layout for the CPU tests (tests/test_sharding.py drives the
same gather over gloo with the oracle as the tracer); the product never calls them."""
import ctypes as C

import numpy as np

from . import _native as N
from . import layouts as L

TILE = 8          # LBVH_TRACE_FAST packet size
GROUP = 8         # adjacent tiles that stay together (lbvh_trace.hip kShardGroup)


def share_items(shard_index, shard_count, width, height):
    """Work items of a share in packed order -> tile index of the full frame (row-major), or -1 for the slots of the last
    group that lie past the frame's last tile (csrc/lbvh_trace.hip shard_tile / shard_work)."""
    tiles_x, tiles_y = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    n_tiles = tiles_x * tiles_y
    groups = (n_tiles + GROUP - 1) // GROUP
    owned = (groups - shard_index + shard_count - 1) // shard_count if groups > shard_index else 0
    out = np.empty(owned * GROUP, dtype=np.int64)
    for k in range(owned * GROUP):
        t = ((k // GROUP) * shard_count + shard_index) * GROUP + k % GROUP
        out[k] = t if t < n_tiles else -1
    return out


def shard_records(width, height, shard_index, shard_count):
    """= lbvh_shard_records"""
    return len(share_items(shard_index, shard_count, width, height)) * TILE * TILE


def pack_share(frame, shard_index, shard_count):
    """(H, W) record array -> the share's packed block (records of lanes outside the screen / slots past the frame stay 0)"""
    height, width = frame.shape
    tiles_x = (width + TILE - 1) // TILE
    items = share_items(shard_index, shard_count, width, height)
    out = np.zeros(len(items) * TILE * TILE, dtype=frame.dtype)
    for k, t in enumerate(items):
        if t < 0:
            continue
        ty, tx = divmod(int(t), tiles_x)
        block = frame[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        dst = out[k * 64:(k + 1) * 64].reshape(TILE, TILE)
        dst[:block.shape[0], :block.shape[1]] = block
    return out


def unpack_shares(packed_shares, shard_count, width, height, frame=None):
    """[share 0's block, share 1's block, ...] -> the (H, W) frame (= lbvh_frame_unpack)"""
    tiles_x = (width + TILE - 1) // TILE
    if frame is None:
        frame = np.zeros((height, width), dtype=packed_shares[0].dtype)
    for s, packed in enumerate(packed_shares):
        for k, t in enumerate(share_items(s, shard_count, width, height)):
            if t < 0:
                continue
            ty, tx = divmod(int(t), tiles_x)
            block = frame[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
            block[:] = packed[k * 64:(k + 1) * 64].reshape(TILE, TILE)[:block.shape[0], :block.shape[1]]
    return frame


class FrameGather:
    """This rank's side of the frame assembly.  `ctx` is the rank's Context (own stream), `dist` an initialised
    torch.distributed, `staged` = the backend moves host memory only (gloo: ranks sharing a GPU in tests)."""

    SELF_TEST_WORDS = 256
    FLAG_SLOTS = 64           # rank r's "my share of frame f is in the buffer" in slot r - 1 ...
    CONSUMED_SLOT = 63        # ... and rank 0's "frame f has been read" in the last one

    def __init__(self, ctx, dist, rank, world, width, height, device_id, staged, mode="auto"):
        from .host import DataBuffer
        import torch
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world
        self.width, self.height, self.device_id, self.staged = width, height, device_id, staged
        self.torch = torch
        self.frame_no = 0
        self.frame = DataBuffer(ctx, width * height, L.HIT) if rank == 0 else None      # the whole frame lives here
        self.mode = None
        self.peer_error = None
        if mode in ("auto", "peer"):
            ok = self._setup_peer()
            if ok:
                self.mode = "peer"
            elif mode == "peer":
                raise RuntimeError(f"peer-mapped frame buffer unavailable: {self.peer_error}")
        if self.mode is None:
            self._setup_packed()
            self.mode = "packed"

    # ---- helpers ----------------------------------------------------------------------------------------------------
    def _all_ok(self, ok):
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int64)
        if not self.staged:
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    # ---- peer-mapped frame buffer -----------------------------------------------------------------------------------
    def _setup_peer(self):
        from .host import DataBuffer
        ctx, dist = self.ctx, self.dist
        self._flags = self._peer_frame = self._peer_flags = None
        payload = [None]
        err = None
        if self.rank == 0:
            try:
                if self.world - 1 > self.CONSUMED_SLOT:
                    raise N.LbvhError(-1, f"the flag array has {self.CONSUMED_SLOT} completion slots")
                self._flags = ctx.flags_alloc(self.FLAG_SLOTS)      # zeroed, uncached (returns after the fill has happened)
                payload = [(self.device_id, ctx.ipc_export(self.frame.device), ctx.ipc_export(self._flags))]
            except N.LbvhError as e:
                err = str(e)
        dist.broadcast_object_list(payload, src=0)
        if payload[0] is None:
            err = err or "rank 0 could not export its frame buffer"
        elif self.rank != 0:
            try:
                owner_device, h_frame, h_flags = payload[0]
                ctx.peer_enable(owner_device)
                self._peer_frame = ctx.ipc_import(h_frame)
                self._peer_flags = ctx.ipc_import(h_flags)
            except N.LbvhError as e:
                err = str(e)
        ok = self._all_ok(err is None)
        if ok:
            # self-test: every rank writes its pattern into rank 0's frame buffer and signals; rank 0 waits on the device
            # and reads the patterns back
            try:
                words = self.SELF_TEST_WORDS
                if self.rank != 0:
                    at = C.c_void_p(self._peer_frame.value + self.rank * words * 4)
                    N.check(ctx.handle, N.lib.lbvh_buffer_fill_u32(ctx.handle, at, 0xA5000000 + self.rank, words))
                    ctx.frame_signal(self._peer_flags, self.rank - 1, 1)
                    ctx.sync()
                else:
                    ctx.frame_wait(self._flags, self.world - 1, 1)
                    got = np.zeros(self.world * words, dtype=np.uint32)
                    N.check(ctx.handle, N.lib.lbvh_buffer_download(ctx.handle, got.ctypes.data_as(C.c_void_p), self.frame.device, got.nbytes))
                    for r in range(1, self.world):
                        if not (got[r * words:(r + 1) * words] == 0xA5000000 + r).all():
                            err = f"self-test: rank {r}'s pattern did not arrive in rank 0's frame buffer"
            except N.LbvhError as e:
                err = str(e)
            ok = self._all_ok(err is None)
        self.frame_no = 1 if ok else 0
        if not ok:
            self.peer_error = err or "another rank could not map rank 0's frame buffer"
            self._close_peer()
        return ok

    def _close_peer(self):
        for p in ("_peer_frame", "_peer_flags"):
            ptr = getattr(self, p, None)
            if ptr is not None:
                try:
                    self.ctx.ipc_close(ptr)
                except N.LbvhError:
                    pass
                setattr(self, p, None)
        if getattr(self, "_flags", None) is not None:
            self.ctx.flags_free(self._flags)
            self._flags = None

    # ---- packed shares + one gather ---------------------------------------------------------------------------------
    def _setup_packed(self):
        torch = self.torch
        self.stride = int(N.lib.lbvh_shard_records(self.width, self.height, 0, self.world))      # the largest share
        dev = torch.device("cuda", self.device_id)
        self._packed = torch.zeros(self.stride * 4, dtype=torch.float32, device=dev)
        self._gathered = torch.zeros(self.world * self.stride * 4, dtype=torch.float32, device=dev) if self.rank == 0 else None
        # the library's stream and torch's (the collective's) are ordered through two events recorded across them
        self._ev_traced = torch.cuda.Event()
        self._ev_gathered = torch.cuda.Event()
        self._ev_traced.record()
        self._ev_gathered.record()
        torch.cuda.synchronize()

    # ---- per frame --------------------------------------------------------------------------------------------------
    def trace_share(self, camera, scene, trace_mode, own_done_event=None, own_start_event=None):
        """Enqueue this rank's share of the frame and its way to rank 0.  On rank 0 the context's stream is, afterwards,
        behind the WHOLE frame (every rank's records are in self.frame for whatever is enqueued next).  own_start_event /
        own_done_event: timing events recorded right in front of / behind this rank's own traversal — behind the wait for
        rank 0's "frame read" word and before any waiting / moving of records — for the `without the gather` figure (the
        span between them holds no cross-rank skew: ADVICE r4)."""
        ctx, rank, world = self.ctx, self.rank, self.world
        self.frame_no += 1
        if self.mode == "peer":
            # frame k may be overwritten only when rank 0 has read it: whatever rank 0 enqueued on its stream since the
            # previous call (shading, a read-back) is in front of this signal, and the others wait for it — on the device, like
            # the completion flags — before their first store of the new frame (the `consumed` event of host.MultiGpuDrawer)
            if rank == 0:
                ctx.frame_signal(self._flags, self.CONSUMED_SLOT, self.frame_no - 1)
            else:
                ctx.frame_wait(C.c_void_p(self._peer_flags.value + 4 * self.CONSUMED_SLOT), 1, self.frame_no - 1)
            if own_start_event is not None:
                ctx.record(own_start_event)
            target = self.frame.device if rank == 0 else self._peer_frame
            N.check(ctx.handle, N.lib.lbvh_trace_primary_shard(ctx.handle, C.byref(camera), rank, world, C.byref(scene), trace_mode, target, None))
            if own_done_event is not None:
                ctx.record(own_done_event)
            if rank == 0:
                ctx.frame_wait(self._flags, world - 1, self.frame_no)
            else:
                ctx.frame_signal(self._peer_flags, rank - 1, self.frame_no)
            return
        torch, dist = self.torch, self.dist
        if own_start_event is not None:
            ctx.record(own_start_event)
        N.check(ctx.handle, N.lib.lbvh_trace_primary_shard_packed(ctx.handle, C.byref(camera), rank, world, C.byref(scene), trace_mode,
                                                                  C.c_void_p(self._packed.data_ptr()), None))
        if own_done_event is not None:
            ctx.record(own_done_event)
        if self.staged:                  # test mode (gloo): through the host
            ctx.sync()
            h = self._packed.cpu()
            parts = [torch.empty_like(h) for _ in range(world)] if rank == 0 else None
            dist.gather(h, parts, dst=0)
            if rank == 0:
                self._gathered.copy_(torch.cat(parts))
                torch.cuda.synchronize()
        else:
            N.check(ctx.handle, N.lib.lbvh_event_record(ctx.handle, C.c_void_p(self._ev_traced.cuda_event)))
            torch.cuda.current_stream().wait_event(self._ev_traced)
            parts = list(self._gathered.view(world, -1).unbind(0)) if rank == 0 else None
            dist.gather(self._packed, parts, dst=0)
            # every rank's library stream goes behind the collective: rank 0 unpacks what it received, and on the others the
            # NEXT frame's trace must not overwrite the packed block while it is still being sent
            self._ev_gathered.record()
            ctx.wait_event(C.c_void_p(self._ev_gathered.cuda_event))
        if rank == 0:
            N.check(ctx.handle, N.lib.lbvh_frame_unpack(ctx.handle, C.c_void_p(self._gathered.data_ptr()), self.stride, 0, world, world,
                                                        self.width, self.height, self.frame.device))

    def close(self):
        self.ctx.sync()
        if self.mode == "peer":
            self.dist.barrier()          # nobody unmaps while another rank may still store ...
            if self.rank != 0:
                self._close_peer()
            self.dist.barrier()          # ... and the owner frees only what nobody maps any more
            if self.rank == 0:
                self._close_peer()
        if self.frame is not None:
            self.frame.dispose()
            self.frame = None
