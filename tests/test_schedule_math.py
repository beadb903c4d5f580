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
