"""Detector sharding across the GPUs of a node.

Every detector row is independent given the (small, replicated) screens,
boresight and tables, so ranks take contiguous detector blocks and regenerate
identical screens from the same Philox key: the data path needs no collective.
``all_gather_tod`` is the optional epilogue the north star names (one all-gather of
the TOD over xGMI, ``nccl`` = RCCL on ROCm; ``gloo`` on CPU tensors in tests).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_det: int, world_size: int, rank: int, align: int = 16):
    """Contiguous block [lo, hi) of rank ``rank``; blocks are multiples of ``align``
    (the upsample kernel's detector tile) except the last, and cover [0, n_det)."""
    per = -(-n_det // world_size)
    per = -(-per // align) * align
    lo = min(rank * per, n_det)
    return lo, min(lo + per, n_det)


def shard_slice(n_det: int, world_size: int = None, rank: int = None, align: int = 16) -> slice:
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return slice(*shard_bounds(n_det, world_size, rank, align))


def all_gather_tod(local: torch.Tensor, n_det: int, align: int = 16, time_chunk: int = None) -> torch.Tensor:
    """Gather the [D_rank, T] shards into the full [n_det, T] TOD on every rank.

    Shards produced by ``shard_slice`` are equal-sized except the last, so the
    gather pads to the common size and trims.  ``time_chunk`` bounds the staging
    buffer (the full gather of config 5 does not fit one GPU: gather chunk by chunk
    and consume each chunk before the next)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    per = shard_bounds(n_det, world, 0, align)[1]
    T = local.shape[1]
    out = torch.empty((n_det, T), dtype=local.dtype, device=local.device)
    step = time_chunk or T
    for s in range(0, T, step):
        e = min(s + step, T)
        send = torch.zeros((per, e - s), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local[:, s:e]
        recv = torch.empty((world * per, e - s), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send)
        out[:, s:e] = recv[:n_det]
    return out


class TodGather:
    """The all-gather of the north star through the C ABI: ``mrx_comm_create`` +
    ``mrx_allgather_tod`` (RCCL over xGMI, loaded by libmrx itself).  The unique id travels
    from rank 0 over the already-initialised ``torch.distributed`` group (any backend) -- the
    only thing torch does here.  ``world == 1`` needs no group at all.

    The TOD is detector-major and shards are equal row blocks, so every rank allocates the
    whole ``[world * rows_per_rank, T]`` array once, the writer puts the shard straight into
    this rank's rows (``my_rows``) and ``gather()`` completes the array in place: no staging."""

    @staticmethod
    def exchange_unique_id(ctx, world: int, rank: int) -> bytes:
        """ncclGetUniqueId on rank 0, handed to every rank over the torch.distributed group (a
        collective: call it from the thread that owns the group)."""
        import ctypes as C

        ident = C.create_string_buffer(128)
        if rank == 0:
            ctx.call("mrx_comm_unique_id", ident)
        box = [ident.raw]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return box[0]

    def __init__(self, ctx, n_det: int, world: int = None, rank: int = None, align: int = 16, unique_id: bytes = None):
        import ctypes as C

        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        self.ctx, self.world, self.rank, self.n_det = ctx, int(world), int(rank), int(n_det)
        self.rows_per_rank = shard_bounds(n_det, world, 0, align)[1] if world > 1 else n_det
        self.lo, self.hi = shard_bounds(n_det, world, rank, align)
        if unique_id is None:
            unique_id = self.exchange_unique_id(ctx, self.world, self.rank)
        ident = C.create_string_buffer(unique_id, 128)
        comm = C.c_void_p()
        ctx.call("mrx_comm_create", ident, self.world, self.rank, C.byref(comm))
        self.comm = comm

    def full_buffer(self, T: int, device) -> torch.Tensor:
        """[world * rows_per_rank, T] float32; rows >= n_det (padding of the last shard) are scratch."""
        return torch.empty((self.world * self.rows_per_rank, T), dtype=torch.float32, device=device)

    def my_rows(self, full: torch.Tensor) -> torch.Tensor:
        return full[self.lo : self.hi]

    def gather(self, full: torch.Tensor, shard: torch.Tensor = None, algo: str = "allgather"):
        """Complete ``full`` on every rank (enqueued on the context's stream).  ``shard``: a
        separate contiguous [rows_per_rank, T] buffer to gather from; default in place.
        ``algo``: "allgather" (one ncclAllGather) or "p2p" (direct sends and receives to every
        peer in one group: mrx_allgather_tod_p2p)."""
        from ._lib import ptr

        assert full.is_contiguous() and full.shape[0] == self.world * self.rows_per_rank
        count = self.rows_per_rank * full.shape[1]
        src = full[self.rank * self.rows_per_rank :] if shard is None else shard
        assert src.is_contiguous() and src.numel() >= count
        name = {"allgather": "mrx_allgather_tod", "p2p": "mrx_allgather_tod_p2p"}[algo]
        self.ctx.call(name, self.comm, ptr(src), ptr(full), count)
        return full[: self.n_det]

    def exchange_screens(self, screens):
        """mrx_exchange_screens: layer l was generated by rank l % world; fill in the others' in place."""
        import ctypes as C

        assert all(t.is_contiguous() and t.dtype == torch.float32 for t in screens)
        ptrs = (C.c_void_p * len(screens))(*[t.data_ptr() for t in screens])
        counts = (C.c_size_t * len(screens))(*[t.numel() for t in screens])
        self.ctx.call("mrx_exchange_screens", self.comm, ptrs, counts, len(screens))
        return screens

    def bytes_received(self, T: int) -> int:
        return (self.world - 1) * self.rows_per_rank * T * 4

    def close(self):
        if getattr(self, "comm", None):
            self.ctx.call("mrx_comm_destroy", self.comm)
            self.comm = None


def layers_of_rank(n_layers: int, world_size: int = None, rank: int = None):
    """Round-robin ownership of the turbulent layers for sharded screen generation."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return list(range(rank, n_layers, world_size))


def exchange_layer_screens(screens):
    """Strong-scaling option, torch.distributed form (any backend; the RCCL form through the C ABI is
    ``TodGather.exchange_screens``): each rank generated only ``layers_of_rank`` into its (persistent,
    plan-bound) screen buffers; one broadcast per layer from its owner fills the rest in
    place.  Screens are functions of (seed, layer) only, so the result is bit for bit what every
    rank would have generated itself.  Layers may differ in shape, hence per-layer broadcasts."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return screens
    world = dist.get_world_size()
    for l, t in enumerate(screens):
        dist.broadcast(t, src=l % world)
    return screens


def stream_gathered_tod(local: torch.Tensor, n_det: int, time_chunk: int, consume=None, align: int = 16) -> int:
    """All-gather the TOD one time chunk at a time into a reusable staging buffer and
    hand each gathered [n_det, chunk] block to ``consume`` (the full gather of a large
    configuration does not fit one GPU; a consumer writes or reduces each block before the
    next arrives).  Returns the number of bytes received per rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if consume is not None:
            consume(0, local)
        return 0
    world = dist.get_world_size()
    per = shard_bounds(n_det, world, 0, align)[1]
    T = local.shape[1]
    send = torch.zeros((per, time_chunk), dtype=local.dtype, device=local.device)
    recv = torch.empty((world * per, time_chunk), dtype=local.dtype, device=local.device)
    received = 0
    for s in range(0, T, time_chunk):
        e = min(s + time_chunk, T)
        if e - s == time_chunk:
            send[: local.shape[0]].copy_(local[:, s:e])
            dist.all_gather_into_tensor(recv, send)
            block = recv[:n_det]
        else:  # ragged tail
            snd = torch.zeros((per, e - s), dtype=local.dtype, device=local.device)
            snd[: local.shape[0]] = local[:, s:e]
            rcv = torch.empty((world * per, e - s), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(rcv, snd)
            block = rcv[:n_det]
        received += (world - 1) * per * (e - s) * local.element_size()
        if consume is not None:
            consume(s, block)
    return received
