"""The index arithmetic of the scheduling options that must not change an image (csrc/hip/render_kernel_impl.h,
stream_kernel_impl.h), restated: every item of a draw is rendered exactly once whatever the pixel order, the lane spread
and the sample split.  (The kernels themselves are compared frame for frame on the GPU: tests/test_gpu_parity.py.)"""
import numpy as np
import pytest


def transposed(q, n_work):
    """render_kernel_impl.h: lane l of the w-th wavefront's worth of items takes item l * (n_work / 64) + w"""
    return (q & 63) * (n_work >> 6) + (q >> 6)


@pytest.mark.parametrize("tiles", [1, 2, 7, 64, 100, 4096, 14400])
def test_transposed_pixel_order_is_a_permutation_of_the_items(tiles):
    n_work = tiles * 64                       # (always whole 8x8 tiles)
    q = np.arange(n_work, dtype=np.int64)
    qs = transposed(q, n_work)
    assert np.array_equal(np.sort(qs), q)
    if tiles >= 64:
        # the 64 lanes of a wavefront hold one pixel of each of 64 different tiles
        first = qs[:64] >> 6
        assert len(set(first.tolist())) == 64


@pytest.mark.parametrize("spread", [1, 2, 4, 16, 64])
@pytest.mark.parametrize("n_items,lanes", [(64, 256), (1000 * 64, 4096), (4096 * 64, 262144), (100, 512)])
def test_lane_spread_and_fixed_lists_cover_every_item_once(spread, n_items, lanes):
    """only every spread-th lane takes items: first item lane / spread, then steps of lanes / spread"""
    stride = lanes // spread
    taken = []
    for lane in range(0, lanes, spread):
        q = lane // spread
        while q < n_items:
            taken.append(q)
            q += stride
    assert sorted(taken) == list(range(n_items))


@pytest.mark.parametrize("split", [1, 2, 8, 32])
def test_split_samples_cover_every_sample_once(split):
    """independent-sample mode: item q = k * n_items + pixel item renders samples k, k + split, ... into plane k"""
    n_items, spp = 192, 40
    seen = np.zeros((n_items, spp), dtype=np.int32)
    for q in range(n_items * split):
        k, item = divmod(q, n_items)
        seen[item, k:spp:split] += 1
    assert (seen == 1).all()


def test_spread_rule_from_the_hit_count():
    """stream_kernel_impl.h: the largest power of two n <= 3 * lanes / expensive pixels, at most 16"""
    def rule(lanes, hits, spp, n_items, split=1):
        expensive = max(1, min(hits // spp, n_items)) * split
        spread = 1
        while spread < 16 and 2 * spread * expensive <= lanes * 3:
            spread *= 2
        return spread
    lanes = 262144
    assert rule(lanes, 175_000 * 256, 256, 921_600) == 4        # dragon/scene.xml, whole frame
    assert rule(lanes, 22_000 * 256, 256, 115_200) == 16        # its 1/8 share
    assert rule(lanes, 700_000 * 512, 512, 1_048_576) == 1      # matpreview, whole frame: dense
    assert rule(lanes, 0, 256, 64) == 16                        # nothing hit: as sparse as allowed


def item_of_pixel(pixel, width, tiles_x, tile_first, tile_stride):
    """path_core.h::item_of_pixel: position of a pixel in a draw's tile enumeration (the pre-pass record, the packed layout)"""
    x, y = pixel % width, pixel // width
    tile = (y >> 3) * tiles_x + (x >> 3)
    return ((tile - tile_first) // tile_stride) * 64 + (y & 7) * 8 + (x & 7)


@pytest.mark.parametrize("width,height", [(64, 64), (203, 117), (1280, 720)])
@pytest.mark.parametrize("rank,world", [(0, 1), (1, 3), (7, 8)])
def test_item_of_pixel_inverts_the_tile_enumeration(width, height, rank, world):
    """item q -> tile tile_first + (q >> 6) * tile_stride, pixel q & 63 of it (render kernels) and back (pre-pass records,
    packed output of the queued renderer): every pixel of a rank's tiles maps to its own item."""
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    tiles = np.arange(rank, tiles_x * tiles_y, world)
    q = np.arange(len(tiles) * 64)
    tile, r = tiles[q >> 6], q & 63
    x, y = (tile % tiles_x) * 8 + (r & 7), (tile // tiles_x) * 8 + (r >> 3)
    inside = (x < width) & (y < height)
    back = item_of_pixel((y * width + x)[inside], width, tiles_x, rank, world)
    assert np.array_equal(back, q[inside])


@pytest.mark.parametrize("n_tiles", [1, 5, 1000])
def test_tile_order_table_is_a_permutation_and_keeps_image_order_within_a_class(n_tiles):
    """hip/tile_order.hip: key = (31 - cost class) << 32 | local tile, sorted ascending -> most expensive class first,
    image order inside a class; hand-out position p renders item (table[p >> 6] << 6) | (p & 63)."""
    rng = np.random.default_rng(n_tiles)
    cls = rng.integers(0, 32, n_tiles)
    keys = ((31 - cls).astype(np.uint64) << np.uint64(32)) | np.arange(n_tiles, dtype=np.uint64)
    table = (np.sort(keys) & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.array_equal(np.sort(table), np.arange(n_tiles))
    assert (np.diff(cls[table]) <= 0).all()                                  # classes descend
    same = np.diff(cls[table]) == 0
    assert (np.diff(table)[same] > 0).all()                                  # image order within a class
    p = np.arange(n_tiles * 64)
    items = (table[p >> 6] << 6) | (p & 63)
    assert np.array_equal(np.sort(items), p)
