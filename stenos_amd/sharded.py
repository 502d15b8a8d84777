"""Superblock-sharded compression across the GPUs of one node (one process per GPU).

Superblocks are independent units in both directions (reference stenos/internal/stenos.cpp:893-904,
1124-1143), so a typed array is cut into contiguous ranges of whole superblocks, one per rank, and
each rank runs the single-GPU codec on its range with no communication.  The only exchange is the
gather of the compressed segments to rank 0 (variable length: sizes by all_gather, payload by direct
point-to-point sends -- with the "nccl" backend that is RCCL over xGMI, one link per peer), which
concatenates them behind one frame header.  The result equals the frame a single GPU produces for the
whole array when the destination buffer is roomy (the reference's capacity rules only differ in the last
superblocks of a tight buffer).

Decoding mirrors it (stenos.cpp:1124-1143, 1151-1202: the reference hands ranges of superblocks to its threads after
walking the [code][csize:3] headers): rank 0 walks the headers of the frame (on its GPU: stenos_hip_frame_index),
cuts the superblock stream at the range boundaries, sends every rank its segment, and each rank decodes its
segment -- behind an 8-byte header of its own -- into its slice of the output.  The slices are gathered only when
the caller wants the whole array in one place.

Frames here are level 0 / 1 frames of the default superblock size (frame byte 0 = 0).
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
    home = local_frame.device
    local_frame = _wire(local_frame, group)
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
        return out.to(home)
    if streams[rank]:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_frame[8:].contiguous(), 0, group)]):
            req.wait()
    return None


def compress_sharded(compress: Callable[[torch.Tensor], torch.Tensor], data: torch.Tensor, bytesoftype: int, group=None, total_bytes: int | None = None):
    """Each rank compresses its own superblock range with `compress` (returns the exact-length frame of a range) and
    rank 0 gets the assembled frame.  data (uint8): with total_bytes given, ONLY this rank's range of the array -- bytes
    shard_ranges(total_bytes, ...)[rank], which is what a rank of a large job holds; without it, the whole array."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if total_bytes is None:
        total_bytes = data.numel()
        begin, end = shard_ranges(total_bytes, bytesoftype, world)[rank]
        data = data[begin:end]
    else:
        begin, end = shard_ranges(total_bytes, bytesoftype, world)[rank]
        assert data.numel() == end - begin, "this rank's slice must be its shard_ranges() range"
    frame = compress(data.contiguous())
    return gather_frames(frame, total_bytes, group)


def _wire(t: torch.Tensor, group=None) -> torch.Tensor:
    """gloo moves host memory only (the CPU test rig and single-GPU rehearsals); nccl / RCCL moves device memory"""
    return t.cpu() if dist.get_backend(group) == "gloo" and t.is_cuda else t


def walk_frame_host(frame: torch.Tensor, bytesoftype: int) -> List[int]:
    """Offsets of the superblock headers of a frame in host memory, and its end (stenos.cpp:1126-1134).  For frames on a
    GPU use Stenos.frame_index (the walk then runs on the device)."""
    f = frame.numpy()
    total = int.from_bytes(f[1:8].tobytes(), "little")
    sb = superblock_bytes(bytesoftype)
    off, out = 8, []
    for _ in range((total + sb - 1) // sb):
        out.append(off)
        off += 4 + int.from_bytes(f[off + 1:off + 4].tobytes(), "little")
    return out + [off]


def segment_table(index: List[int], total_bytes: int, bytesoftype: int, world: int) -> torch.Tensor:
    """Row r: [first frame byte, end frame byte, first output byte, end output byte] of rank r's superblock range."""
    sb = superblock_bytes(bytesoftype)
    rows = []
    for begin, end in shard_ranges(total_bytes, bytesoftype, world):
        s0, s1 = begin // sb, (end + sb - 1) // sb
        rows.append([index[s0], index[s1], begin, end])
    return torch.tensor(rows, dtype=torch.int64)


def scatter_segments(frame: torch.Tensor | None, table: torch.Tensor | None, device, group=None):
    """Rank 0 passes the frame and its segment_table; every rank gets back (the frame of its own range: an 8-byte
    header + its segment of the superblock stream, first output byte, end output byte)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    table = table.clone() if rank == 0 else torch.zeros((world, 4), dtype=torch.int64)
    wire_dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else device
    table = table.to(wire_dev)
    dist.broadcast(table, 0, group=group)
    rows = table.cpu().tolist()
    f0, f1, o0, o1 = rows[rank]
    local = torch.empty(8 + f1 - f0, dtype=torch.uint8, device=wire_dev)
    local[:8] = torch.tensor([0] + [((o1 - o0) >> (8 * i)) & 0xFF for i in range(7)], dtype=torch.uint8, device=wire_dev)
    if rank == 0:
        src = _wire(frame, group)
        local[8:] = src[f0:f1]
        ops = [dist.P2POp(dist.isend, src[a:b].contiguous(), r, group) for r, (a, b, _, _) in enumerate(rows) if r and b > a]
    else:
        ops = [dist.P2POp(dist.irecv, local[8:], 0, group)] if f1 > f0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return local.to(device), o0, o1


def decompress_sharded(decompress: Callable[[torch.Tensor, int], torch.Tensor], frame: torch.Tensor | None, index: List[int] | None, total_bytes: int,
                       bytesoftype: int, device, group=None, gather_output: bool = False):
    """Rank 0 passes the whole frame and the offsets of its superblock headers (Stenos.frame_index / walk_frame_host),
    the other ranks None.  `decompress(frame_of_a_range, decoded_bytes)` returns the decoded bytes of a range on
    `device`.  Every rank returns (its slice of the output, first byte, end byte); with gather_output rank 0 returns the
    whole array instead of its slice."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    table = segment_table(index, total_bytes, bytesoftype, world) if rank == 0 else None
    local, o0, o1 = scatter_segments(frame, table, device, group)
    part = decompress(local, o1 - o0) if o1 > o0 else torch.empty(0, dtype=torch.uint8, device=device)
    if not gather_output:
        return part, o0, o1
    ranges = shard_ranges(total_bytes, bytesoftype, world)
    if rank == 0:
        out = torch.empty(total_bytes, dtype=torch.uint8, device=_wire(part, group).device)
        out[o0:o1] = _wire(part, group)
        ops = [dist.P2POp(dist.irecv, out[a:b], r, group) for r, (a, b) in enumerate(ranges) if r and b > a]
    else:
        out = None
        ops = [dist.P2POp(dist.isend, _wire(part, group).contiguous(), 0, group)] if o1 > o0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return (out.to(device) if out is not None else None), o0, o1
