"""Multi-GPU partition of a frame: interleaved 8-row scanline tiles, one process per GPU.

Tile k of the frame (rows [8k, 8k+8)) belongs to rank k mod G.  Every rank renders its tiles
with identical arithmetic into a compact buffer [n_local_tiles*8][W][C]; the buffers are
gathered to rank 0 (RCCL over xGMI with the "nccl" backend, gloo on CPU) and de-interleaved
there.  Pixels never interact and the scene is replicated, so there is no other exchange.
The reference has no counterpart (single process, single device).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

TILE_ROWS = 8


def total_tiles(height: int) -> int:
    return (height + TILE_ROWS - 1) // TILE_ROWS


def local_tiles(height: int, rank: int, world: int) -> int:
    """Number of tiles rank `rank` of `world` renders (tile_first = rank, tile_step = world)."""
    t = total_tiles(height)
    return 0 if rank >= t else (t - rank + world - 1) // world


def max_local_tiles(height: int, world: int) -> int:
    return local_tiles(height, 0, world)


def deinterleave(gathered: torch.Tensor, height: int) -> torch.Tensor:
    """gathered: [G][T*8][W][C] with rank r's local tile j = frame tile j*G + r.  -> [height][W][C]."""
    G, rows, W, Cc = gathered.shape
    T = rows // TILE_ROWS
    x = gathered.reshape(G, T, TILE_ROWS, W, Cc).permute(1, 0, 2, 3, 4).reshape(T * G * TILE_ROWS, W, Cc)
    return x[:height].contiguous()


def gather_buffer(local: torch.Tensor, group=None, dst: int = 0):
    """The [G][T_max*8][W][C] buffer rank `dst` receives the tile buffers in (None on the other ranks): allocated once
    per output slot by a render loop and handed to every gather_frame of that slot."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 or dist.get_rank(group) != dst:
        return None
    return torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)


def gather_frame(local: torch.Tensor, height: int, group=None, dst: int = 0, buf: torch.Tensor | None = None):
    """Gather every rank's compact tile buffer ([T_max*8][W][C], same shape on all ranks) to
    rank `dst` and return the assembled frame there (None elsewhere).  `buf`: rank dst's receive buffer
    (gather_buffer); without one a buffer is allocated for this call."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return deinterleave(local.unsqueeze(0), height)
    rank = dist.get_rank(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo gathers host tensors only: device buffers go through host memory (the rehearsal of an N > 1 run on a box
        # with one GPU, all ranks on that GPU; the real path is RCCL and never comes here)
        frame = gather_frame(local.cpu(), height, group, dst)
        return frame.to(local.device) if frame is not None else None
    if rank == dst:
        if buf is None:
            buf = gather_buffer(local, group, dst)
        dist.gather(local, gather_list=[buf[i] for i in range(world)], dst=dst, group=group)
        return deinterleave(buf, height)
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None
