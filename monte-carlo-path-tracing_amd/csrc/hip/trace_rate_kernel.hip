// Experiment kernel (mcpt_debug_trace_rate): how fast can a LEAN kernel — walk state only, no path state, no shading
// code — answer closest-hit queries for a batch of incoherent rays, (a) one ray per lane, a wavefront lasting as long as
// its slowest ray, or (b) persistent wavefronts that re-fill free lanes from a global queue (stream_core.h::stream_trace
// over rays in HBM), at 4 or 8 wavefronts per SIMD?  It answers what a multi-kernel wavefront formulation could gain on
// meshes, where the render kernels are bound by the latency of one node fetch per wavefront at a time (DESIGN.md
// sections 6, 9).  Not used by any render path.
#include <hip/hip_runtime.h>

#include "../stream_core.h"
#include "../short_stack.h"
#include "render_kernel.h"

namespace mcpt
{

namespace
{

constexpr uint32_t kLeanFeatures = kFeatOrderedWalk | kFeatVoteWalk | kFeatSlivers; // triangles only

template <int kWaves>
__global__ void __launch_bounds__(kBlockSize, kWaves) one_ray_per_lane(const DeviceScene sc, uint32_t n, const float *__restrict__ rays,
                                                                      uint32_t *__restrict__ found)
{
    extern __shared__ uint32_t lds_stack[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    Ray ray = make_ray(V3{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, V3{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
    HitRaw raw;
    TraceStats ts{0, 0, 0, 0};
    const bool hit = walk_ordered_vote<false, false, false, true>(sc, lds_stack + threadIdx.x, ray, raw, ts);
    found[i] = hit ? raw.prim : kNone;
}

// mode 2: one ray per lane with the SHORT stack (short_stack.h: kRing entries per lane in LDS, older ones in `spill`)
template <uint32_t kRing>
__global__ void __launch_bounds__(kBlockSize, 8) one_ray_per_lane_short(const DeviceScene sc, uint32_t n, const float *__restrict__ rays,
                                                                        uint32_t *__restrict__ found, uint32_t *__restrict__ spill)
{
    __shared__ uint32_t lds_rings[kRing * kBlockSize];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    ShortStack<kRing> stack{lds_rings + threadIdx.x, spill + i, gridDim.x * kBlockSize, 0u};
    Ray ray = make_ray(V3{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, V3{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
    HitRaw raw;
    const bool hit = walk_ordered_short<false, false, true, kRing>(sc, stack, ray, raw);
    found[i] = hit ? raw.prim : kNone;
}

// mode 3: one ray per lane on the 4-wide quantised hierarchy (short_stack.h: walk_wide_vote; kWideRing entries per lane in LDS)
template <int kWaves>
__global__ void __launch_bounds__(kBlockSize, kWaves) one_ray_per_lane_wide(const DeviceScene sc, uint32_t n, const float *__restrict__ rays,
                                                                           uint32_t *__restrict__ found)
{
    __shared__ uint32_t lds_rings[kWideRing * kBlockSize];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    Ray ray = make_ray(V3{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, V3{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
    HitRaw raw;
    TraceStats ts{0, 0, 0, 0};
    const bool hit = walk_wide_vote<false, false, false, true>(sc, lds_rings + threadIdx.x, ray, raw, ts);
    found[i] = hit ? raw.prim : kNone;
}

template <int kWaves>
__global__ void __launch_bounds__(kBlockSize, kWaves) refill_from_queue(const DeviceScene sc, uint32_t n, uint32_t *__restrict__ hot,
                                                                       const uint32_t *__restrict__ ids, uint32_t *__restrict__ next,
                                                                       uint32_t refill_at)
{
    extern __shared__ uint32_t lds_stack[];
    StreamStore m;
    m.hot = hot, m.cold = nullptr, m.P = n;
    const StreamRayList list{ids, n, 0u, next};
    stream_trace<Config<kLeanFeatures>, false>(sc, m, list, lds_stack + threadIdx.x, refill_at, nullptr, nullptr);
}

__global__ void fill_queue(uint32_t n, const float *__restrict__ rays, uint32_t *__restrict__ hot, uint32_t *__restrict__ ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    ids[i] = i;
    for (uint32_t k = 0; k < 3; ++k)
    {
        hot[(kHotA + k) * static_cast<size_t>(n) + i] = __float_as_uint(rays[6 * i + k]);
        hot[(kHotDir + k) * static_cast<size_t>(n) + i] = __float_as_uint(rays[6 * i + 3 + k]);
    }
}

} // namespace

// mode 0: one ray per lane; mode 1: persistent wavefronts re-filling from the queue; mode 2: one ray per lane with the
// short traversal stack (8 wavefronts per SIMD; refill_at = ring size 4 / 8 / 16).  waves: 4 or 8 per SIMD.
// found[i] = primitive hit by ray i (kNone: none); milliseconds = the trace kernel alone (HIP events).
hipError_t RunTraceRate(const DeviceScene &sc, uint32_t n, const float *rays_dev, int mode, int waves, uint32_t refill_at, uint32_t n_cus,
                        uint32_t *found_dev, float *milliseconds, hipStream_t stream)
{
    const size_t lds_bytes = size_t(sc.integrator.walk_depth) * kBlockSize * sizeof(uint32_t);
    hipEvent_t e0, e1;
    hipError_t err = hipEventCreate(&e0);
    if (err == hipSuccess)
        err = hipEventCreate(&e1);
    if (err != hipSuccess)
        return err;
    uint32_t *hot = nullptr, *ids = nullptr, *next = nullptr;
    if (mode == 1)
    {
        if ((err = hipMalloc(reinterpret_cast<void **>(&hot), size_t(stream_hot_words(1)) * n * 4)) != hipSuccess ||
            (err = hipMalloc(reinterpret_cast<void **>(&ids), size_t(2) * n * 4)) != hipSuccess ||
            (err = hipMalloc(reinterpret_cast<void **>(&next), 4)) != hipSuccess)
            return err;
        (void)hipMemsetAsync(next, 0, 4, stream);
        hipLaunchKernelGGL(fill_queue, dim3((n + 255) / 256), dim3(256), 0, stream, n, rays_dev, hot, ids);
    }
    uint32_t *spill = nullptr;
    if (mode == 2)
    {
        const size_t words = size_t(sc.integrator.walk_depth) * ((n + 255) / 256) * 256;
        if ((err = hipMalloc(reinterpret_cast<void **>(&spill), words * 4)) != hipSuccess)
            return err;
    }
    (void)hipEventRecord(e0, stream);
    if (mode == 3)
    {
        if (waves >= 8)
            hipLaunchKernelGGL(one_ray_per_lane_wide<8>, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, rays_dev, found_dev);
        else
            hipLaunchKernelGGL(one_ray_per_lane_wide<4>, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, rays_dev, found_dev);
    }
    else if (mode == 2)
    {
        // refill_at selects the ring size here: 4, 8 (default) or 16 entries per lane in LDS
        if (refill_at == 4)
            hipLaunchKernelGGL(one_ray_per_lane_short<4>, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, rays_dev, found_dev, spill);
        else if (refill_at == 16)
            hipLaunchKernelGGL(one_ray_per_lane_short<16>, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, rays_dev, found_dev, spill);
        else
            hipLaunchKernelGGL(one_ray_per_lane_short<8>, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, rays_dev, found_dev, spill);
    }
    else if (mode == 0)
    {
        if (waves >= 8)
            hipLaunchKernelGGL(one_ray_per_lane<8>, dim3((n + 255) / 256), dim3(256), lds_bytes, stream, sc, n, rays_dev, found_dev);
        else
            hipLaunchKernelGGL(one_ray_per_lane<4>, dim3((n + 255) / 256), dim3(256), lds_bytes, stream, sc, n, rays_dev, found_dev);
    }
    else
    {
        int per_cu = 0;
        if (waves >= 8)
            err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, refill_from_queue<8>, kBlockSize, lds_bytes);
        else
            err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, refill_from_queue<4>, kBlockSize, lds_bytes);
        if (err != hipSuccess)
            return err;
        const uint32_t max_per_cu = static_cast<uint32_t>(waves >= 8 ? 8 : 4);
        const uint32_t blocks = n_cus * std::min(static_cast<uint32_t>(per_cu < 1 ? 1 : per_cu), max_per_cu);
        if (waves >= 8)
            hipLaunchKernelGGL(refill_from_queue<8>, dim3(blocks), dim3(256), lds_bytes, stream, sc, n, hot, ids, next, refill_at);
        else
            hipLaunchKernelGGL(refill_from_queue<4>, dim3(blocks), dim3(256), lds_bytes, stream, sc, n, hot, ids, next, refill_at);
    }
    (void)hipEventRecord(e1, stream);
    err = hipStreamSynchronize(stream);
    if (err == hipSuccess)
        err = hipEventElapsedTime(milliseconds, e0, e1);
    if (mode == 1 && err == hipSuccess)
        err = hipMemcpyAsync(found_dev, hot + size_t(kHotPrim) * n, size_t(n) * 4, hipMemcpyDeviceToDevice, stream);
    (void)hipStreamSynchronize(stream);
    (void)hipFree(hot), (void)hipFree(ids), (void)hipFree(next), (void)hipFree(spill);
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
    return err;
}

} // namespace mcpt
