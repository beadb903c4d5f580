// Cost-ordered tile hand-out for gfx950 (MI355X): which tiles the work counter should hand out first.
//
// The reference threads one random stream through all samples of a pixel (src/renderer/renderer.cpp:62-81), so a pixel
// is a sequential chain of rounds and a frame ends with the chains that started LAST.  The work counter hands the tiles
// out in image order; if an expensive region sits at the end of that order the GPU drains while a few long chains
// finish (round 2 measured both signs of it by reversing the order: dining-room +25 %, matpreview rough conductor
// -22 %).  The pre-pass (primary_kernel.hip) already knows what every camera ray hits, so a tile's cost can be estimated
// before the chains start: per pixel, the first few samples' camera-ray hits weighted by the BSDF kind they land on
// (a camera ray that leaves the scene ends its sample at once; a path that enters a dielectric bounces for long).
// One wavefront per tile sums its 64 pixels; the tiles are then sorted by cost class, most expensive first, image order
// within a class (rocPRIM radix sort of 64-bit keys: ~16 k keys, microseconds), and the render kernels map hand-out
// position -> tile through the sorted table (RenderJob::tile_order).  The image does not depend on it.
// No reference counterpart (its CPU back end hands out Morton-ordered 64-pixel patches, renderer.cpp:24-60).
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "../path_core.h"
#include "render_kernel.h"

namespace mcpt
{

namespace
{

// relative length of the chain a first vertex of this kind starts (measured rounds per sample on the BASELINE scenes:
// diffuse walls 2-3, conductor 3-4, dielectric 8-9)
__device__ __forceinline__ uint32_t kind_weight(const DeviceScene &sc, uint32_t inst)
{
    const uint32_t b = sc.instances[inst].bsdf;
    if (b == kNone)
        return 4u;
    switch (sc.bsdfs[b].kind)
    {
    case kBsdfAreaLight: return 1u;
    case kBsdfConductor: return 6u;
    case kBsdfPlastic: return 6u;
    case kBsdfThinDielectric: return 8u;
    case kBsdfDielectric: return 12u;
    default: return 4u;
    }
}

constexpr uint32_t kCostSamples = 4; // camera rays per pixel that enter the estimate

__global__ void __launch_bounds__(kBlockSize) tile_cost_kernel(const DeviceScene sc, const RenderJob job, const uint32_t *__restrict__ prehit,
                                                               unsigned long long *__restrict__ keys, uint32_t n_tiles)
{
    const uint32_t local_tile = (blockIdx.x * kBlockSize + threadIdx.x) >> 6, r = threadIdx.x & 63u;
    if (local_tile >= n_tiles)
        return;
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height), spp = sc.camera.spp;
    const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
    const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
    uint32_t cost = 0;
    if (x < width && y < height)
    {
        const size_t item = static_cast<size_t>(local_tile) * 64u + r;
        const uint32_t n = spp < kCostSamples ? spp : kCostSamples;
        for (uint32_t s = 0; s < n; ++s)
        {
            // (spread over the pixel's samples: s * spp / n)
            const uint32_t *rec = prehit + 2 * (item * spp + static_cast<size_t>(s) * spp / n);
            cost += rec[0] == kNone ? 1u : kind_weight(sc, rec[1]);
        }
    }
    cost = lanes_sum(cost);
    // Only for jobs whose camera rays mostly hit something (interiors, close-ups).  Where most of them leave the scene
    // (dragon/scene.xml: 81 %) the few expensive pixels are better left interleaved with the cheap ones: the stream
    // kernel's lane spread (stream_kernel_impl.h) already gives every expensive chain a wavefront slot of its own, and
    // bunching them at the front of the order puts them side by side in the same wavefronts — measured 1125 -> 1050
    // Msamples/s on dragon against 374 -> 413 on matpreview rough dielectric.
    bool sorted_order = true;
    if (job.hit_counters)
    {
        unsigned long long hits = 0;
        for (uint32_t k = 0; k < kHitCounters; ++k)
            hits += job.hit_counters[k];
        sorted_order = 2ull * hits >= static_cast<unsigned long long>(job.n_items) * spp;
    }
    if (r == 0 && !sorted_order)
        keys[local_tile] = local_tile; // image order
    if (r == 0 && sorted_order)
    {
        // cost CLASS (32 classes over the possible range), most expensive first; image order within a class, so that
        // tiles handed out together still lie together (locality: DESIGN.md section 3c)
        const uint32_t cls = cost * 31u / (64u * kCostSamples * 12u);
        keys[local_tile] = (static_cast<unsigned long long>(31u - (cls > 31u ? 31u : cls)) << 32) | local_tile;
    }
}

} // namespace

namespace
{
__global__ void keys_from_steps_kernel(const uint32_t *__restrict__ steps, unsigned long long *__restrict__ keys, uint32_t n_tiles)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles)
        keys[t] = (static_cast<unsigned long long>(0xFFFFFFFFu - steps[t]) << 32) | t;
}
} // namespace

hipError_t LaunchTileOrderFromSteps(const uint32_t *steps, uint32_t n_tiles, unsigned long long *keys, unsigned long long *sorted,
                                    void *temp, size_t temp_bytes, hipStream_t stream)
{
    if (n_tiles == 0)
        return hipSuccess;
    hipLaunchKernelGGL(keys_from_steps_kernel, dim3((n_tiles + kBlockSize - 1) / kBlockSize), dim3(kBlockSize), 0, stream, steps, keys, n_tiles);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess)
        return err;
    return rocprim::radix_sort_keys(temp, temp_bytes, keys, sorted, n_tiles, 0, 64, stream);
}

size_t TileOrderTempBytes(uint32_t n_tiles)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, static_cast<unsigned long long *>(nullptr), static_cast<unsigned long long *>(nullptr), n_tiles);
    return bytes ? bytes : 16;
}

hipError_t LaunchTileOrder(const DeviceScene &sc, const RenderJob &job, const uint32_t *prehit, unsigned long long *keys,
                           unsigned long long *sorted, void *temp, size_t temp_bytes, hipStream_t stream)
{
    const uint32_t n_tiles = job.n_items / 64u;
    if (n_tiles == 0)
        return hipSuccess;
    hipLaunchKernelGGL(tile_cost_kernel, dim3((n_tiles * 64u + kBlockSize - 1) / kBlockSize), dim3(kBlockSize), 0, stream, sc, job, prehit, keys,
                       n_tiles);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess)
        return err;
    return rocprim::radix_sort_keys(temp, temp_bytes, keys, sorted, n_tiles, 0, 37, stream);
}

} // namespace mcpt
