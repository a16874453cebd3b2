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
