"""The N>1 path on CPU: two gloo processes each produce the pixels of their
own tile set (with the oracle as the pixel source), pack them the way the HIP
kernel packs tiles, gather once to rank 0 and scatter into the frame.  The
assembled frame must equal the single-process frame bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mcsd_path, width, height, out_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import torch
    import torch.distributed as dist
    import checkers
    from _pkg import load_package
    pkg = load_package()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fg = pkg.tiling.FrameGather(world, rank, width, height, torch.device("cpu"))
    # "render" this rank's tiles: full oracle frame, then keep only own pixels
    full, _ = checkers.Oracle().render(mcsd_path, threads=2)
    tiles = pkg.tiling.rank_tiles(rank, world, width, height)
    x, y, ok = pkg.tiling.tile_pixel_grid(tiles, width, height)
    packed = np.full((fg.max_tiles, 64, 3), -1.0, dtype=np.float32)   # poison padding
    packed[:len(tiles)][ok] = full[y[ok], x[ok]]
    fg.packed.copy_(torch.from_numpy(packed.reshape(-1)))
    frame = fg.gather()
    if rank == 0:
        np.save(out_path, frame.numpy().reshape(height, width, 3))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_of_tiles_rebuilds_the_frame(world, pkg, oracle, mcsd_file, tmp_path):
    import torch.multiprocessing as mp
    width, height = 44, 29  # partial edge tiles in both directions
    scene = pkg.scenes.cornell_box(width, height, 2)
    path = mcsd_file(scene)
    want, _ = oracle.render(path)
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, path, width, height, out), nprocs=world, join=True)
    got = np.load(out)
    assert np.array_equal(got, want)


def test_partition_covers_every_pixel_once(pkg):
    for world in (1, 2, 3, 8):
        for (w, h) in ((512, 512), (1280, 720), (21, 13)):
            src, dst = pkg.tiling.scatter_map(world, w, h)
            assert len(dst) == w * h and len(np.unique(dst)) == w * h
            assert len(np.unique(src)) == len(src)
            counts = [len(pkg.tiling.rank_tiles(r, world, w, h)) for r in range(world)]
            assert max(counts) - min(counts) <= 1 and max(counts) == pkg.tiling.max_tiles_per_rank(world, w, h)
