"""Image-space partition of one frame over the GPUs of a node.

The frame is cut into 8x8-pixel tiles in row-major tile order; rank r of N
owns tiles r, r+N, r+2N, ... (round-robin keeps expensive image regions spread
over all ranks).  Pixels are independent units of work — the reference seeds
one RNG stream per pixel (src/renderer/renderer.cpp:62-66) and applies no
inter-pixel filter — so the partition changes no pixel value.  Each rank renders
its tiles into a packed buffer (64 pixels x 3 floats per tile, padded to the
largest per-rank tile count so that all blocks are the same size), ONE gather
collects the blocks on rank 0 (RCCL over xGMI on the GPU box, gloo in the CPU
tests), and rank 0 scatters them into the frame with a precomputed index map.
"""
from __future__ import annotations

import numpy as np

TILE = 8


def tiles_xy(width: int, height: int):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def rank_tiles(rank: int, world: int, width: int, height: int) -> np.ndarray:
    tx, ty = tiles_xy(width, height)
    return np.arange(rank, tx * ty, world)


def max_tiles_per_rank(world: int, width: int, height: int) -> int:
    return len(rank_tiles(0, world, width, height))


def tile_pixel_grid(tiles: np.ndarray, width: int, height: int):
    """For tiles[k] and in-tile slot r (0..63): pixel x, y and validity mask."""
    tx, _ = tiles_xy(width, height)
    t = tiles[:, None]
    r = np.arange(TILE * TILE)[None, :]
    x = (t % tx) * TILE + r % TILE
    y = (t // tx) * TILE + r // TILE
    return x, y, (x < width) & (y < height)


def scatter_map(world: int, width: int, height: int):
    """(src, dst): concatenated-gather slot index -> frame pixel index."""
    max_tiles = max_tiles_per_rank(world, width, height)
    src, dst = [], []
    for r in range(world):
        tiles = rank_tiles(r, world, width, height)
        x, y, ok = tile_pixel_grid(tiles, width, height)
        slot = (r * max_tiles + np.arange(len(tiles))[:, None]) * (TILE * TILE) + np.arange(TILE * TILE)[None, :]
        src.append(slot[ok])
        dst.append((y * width + x)[ok])
    return np.concatenate(src).astype(np.int64), np.concatenate(dst).astype(np.int64)


class FrameGather:
    """One gather of finished tiles to rank 0 + scatter into the frame."""

    def __init__(self, world: int, rank: int, width: int, height: int, device):
        import torch
        self.world, self.rank, self.width, self.height = world, rank, width, height
        self.max_tiles = max_tiles_per_rank(world, width, height)
        self.packed = torch.zeros(self.max_tiles * TILE * TILE * 3, dtype=torch.float32, device=device)
        self.frame = torch.zeros(height * width * 3, dtype=torch.float32, device=device) if rank == 0 else None
        self.blocks = [torch.empty_like(self.packed) for _ in range(world)] if rank == 0 else None
        if rank == 0:
            src, dst = scatter_map(world, width, height)
            self.src = torch.from_numpy(src).to(device)
            self.dst = torch.from_numpy(dst).to(device)

    def gather(self):
        """Collective: every rank contributes self.packed; rank 0 gets the frame."""
        import torch
        import torch.distributed as dist
        dist.gather(self.packed, self.blocks, dst=0)
        if self.rank == 0:
            self.frame.view(-1, 3)[self.dst] = torch.cat(self.blocks).view(-1, 3)[self.src]
        return self.frame


def communicator_report(device):
    """What the process group itself says about the run: its size counted BY the collective (an all-reduce of ones — not
    WORLD_SIZE from the environment), the backend, and this rank's number."""
    import torch
    import torch.distributed as dist
    ones = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(ones)
    return {"ranks": int(round(float(ones.item()))), "backend": dist.get_backend(), "group_size": dist.get_world_size()}


def self_check(fg: "FrameGather", single_frame):
    """Rank 0: the frame the gather assembled against the frame ONE renderer draws alone (`single_frame`: H x W x 3 float32, or
    None on the other ranks) — sha256 of both and whether they are equal, bit for bit.  A tiling, packing or exchange error
    cannot hide behind a plausible throughput number."""
    import hashlib
    import numpy as np
    if fg.rank != 0:
        return None
    got = fg.frame.detach().cpu().numpy().reshape(fg.height, fg.width, 3)
    want = np.ascontiguousarray(single_frame, dtype=np.float32)
    return {"gathered_frame_sha256": hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest(),
            "single_gpu_frame_sha256": hashlib.sha256(want.tobytes()).hexdigest(),
            "equals_single_gpu_frame": bool(np.array_equal(got, want))}
