"""Superblock-sharded compression across the GPUs of one node (one process per GPU).

Superblocks are independent units in both directions (reference stenos/internal/stenos.cpp:893-904,
1124-1143), so a typed array is cut into contiguous ranges of whole superblocks, one per rank, and
each rank runs the single-GPU codec on its range with no communication.  The only exchange is the
gather of the compressed segments to rank 0 (variable length: sizes by all_gather, payload by direct
point-to-point sends -- with the "nccl" backend that is RCCL over xGMI, one link per peer), which
concatenates them behind one frame header.  The result equals the frame a single GPU produces for the
whole array when the destination buffer is roomy (the reference's capacity rules only differ in the last
superblocks of a tight buffer).
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist

SB_DEFAULT = 131072


def superblock_bytes(bytesoftype: int) -> int:
    """stenos.cpp:71-76 (level 1: shift 0)"""
    bs = 256 * bytesoftype
    return bs if bs > SB_DEFAULT else (SB_DEFAULT // bs) * bs


def shard_ranges(total_bytes: int, bytesoftype: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous byte ranges, one per rank, cut at superblock boundaries."""
    sb = superblock_bytes(bytesoftype)
    nsb = (total_bytes + sb - 1) // sb
    per, extra = divmod(nsb, world)
    out, begin = [], 0
    for r in range(world):
        n = per + (1 if r < extra else 0)
        end = min(total_bytes, begin + n * sb)
        out.append((begin, end))
        begin = end
    return out


def gather_frames(local_frame: torch.Tensor, total_bytes: int, group=None) -> torch.Tensor | None:
    """Every rank passes the frame of its own range (uint8 tensor, exact length).  Rank 0 returns the
    frame of the whole array: [shift 0][total_bytes:7] followed by every rank's superblock stream."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = local_frame.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local_frame.numel()], dtype=torch.int64, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    streams = [s - 8 for s in sizes]  # without each shard's own 8-byte frame header
    if rank == 0:
        out = torch.empty(8 + sum(streams), dtype=torch.uint8, device=dev)
        hdr = [0] + [(total_bytes >> (8 * i)) & 0xFF for i in range(7)]
        out[:8] = torch.tensor(hdr, dtype=torch.uint8, device=dev)
        out[8:8 + streams[0]] = local_frame[8:]
        ops, off = [], 8 + streams[0]
        for r in range(1, world):
            if streams[r]:
                ops.append(dist.P2POp(dist.irecv, out[off:off + streams[r]], r, group))
            off += streams[r]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    if streams[rank]:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_frame[8:].contiguous(), 0, group)]):
            req.wait()
    return None


def compress_sharded(compress: Callable[[torch.Tensor], torch.Tensor], data: torch.Tensor, bytesoftype: int, group=None):
    """data: this rank's view of the WHOLE array (uint8).  Each rank compresses its own superblock range
    with `compress` (returns the exact-length frame of a range) and rank 0 gets the assembled frame."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    begin, end = shard_ranges(data.numel(), bytesoftype, world)[rank]
    frame = compress(data[begin:end].contiguous())
    return gather_frames(frame, data.numel(), group)
