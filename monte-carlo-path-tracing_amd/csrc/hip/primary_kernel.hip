// Primary-visibility pre-pass for gfx950 (MI355X): the closest hit of EVERY camera ray of a frame, ahead of the
// per-pixel sample chains.
//
// Why it exists.  The reference threads one random stream through all samples of a pixel (renderer.cpp:62-81), so a
// pixel's samples are one sequential chain and the render kernels are built around that.  But the CAMERA ray of sample
// s of pixel p does not depend on the stream: its jitter is stratified in x and a van der Corput point in y
// (renderer.cpp:68-76), a function of (p, s) alone, and the ordered walk draws no random numbers.  Its closest hit can
// therefore be computed for all W x H x spp camera rays at once — one lane per (pixel, sample), 64 consecutive samples
// of one pixel per wavefront (rays a fraction of a pixel apart: the wavefront walks the hierarchy in lock step, the
// loads hit L1), a lean kernel (walk state only, no path state, no shading code).  The render kernels then start every
// sample at its first vertex (path_core.h::path_step, stream_core.h::stream_shade) and re-evaluate only the winner's own
// test (traversal.h::hit_from_record) — same hit record, same image — so 1 of the 1.2 (dragon) ... 5.2 (cornell)
// closest rays per sample leaves the sequential chain, and a sample whose camera ray misses costs the chain a few
// dozen instructions.
//
// Output: two words per (pixel, sample) at 2 (item * spp + s), item = the pixel's position in the draw's tile enumeration:
// primitive (kNone: miss), instance.
// Memory: 8 B per sample of the DRAW (a rank's tile share of an N-GPU frame: 1 / N of it) in HBM (dragon 1280x720 spp
// 256, whole frame: 1.9 GB; sized for 288 GB).
// Replaces nothing in the reference one to one: it is the camera-ray part of ShadePath's first Scene::Intersect
// (src/renderer/integrators/path.cpp:18-21) hoisted out of the per-pixel loop.
#include <hip/hip_runtime.h>

#include "../path_core.h"
#include "render_kernel.h"

namespace mcpt
{

namespace
{

template <bool kAnalytic, bool kSlivers, bool kVote, bool kCount>
__global__ void __launch_bounds__(kBlockSize) primary_kernel(const DeviceScene sc, const RenderJob job, uint32_t *__restrict__ prehit,
                                                             TraceCounters *__restrict__ counters)
{
    extern __shared__ uint32_t lds_stacks[];
    uint32_t *stack = lds_stacks + threadIdx.x;
    const uint32_t spp = sc.camera.spp, width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);
    const unsigned long long total = static_cast<unsigned long long>(job.n_items) * spp;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
    TraceStats ts{0, 0, 0, 0};
    uint32_t rays = 0, hits = 0;
    __shared__ uint32_t block_hits;
    if (threadIdx.x == 0)
        block_hits = 0;
    __syncthreads();
    for (unsigned long long q = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; q < total; q += stride)
    {
        const uint32_t item = static_cast<uint32_t>(q / spp), s = static_cast<uint32_t>(q - static_cast<unsigned long long>(item) * spp);
        const uint32_t local_tile = item >> 6, r = item & 63u;
        const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
        const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
        if (x >= width || y >= height)
            continue; // padding of an edge tile
        PathState st;
        st.pixel = y * width + x, st.sample = s;
        start_sample(sc, st); // the reference's camera ray of this (pixel, sample)
        Ray ray = make_ray(st.origin, st.dir);
        HitRaw hit;
        const bool found = kVote ? walk_ordered_vote<false, kAnalytic, kCount, kSlivers>(sc, stack, ray, hit, ts)
                                 : walk_ordered<false, kAnalytic, kCount, kSlivers>(sc, stack, ray, hit, ts);
        uint32_t *rec = prehit + 2 * (static_cast<size_t>(item) * spp + s); // (by work item of THIS draw: path_core.h::prehit_record)
        rec[0] = found ? hit.prim : kNone, rec[1] = found ? hit.inst : 0u;
        ++rays;
        hits += found ? 1u : 0u;
    }
    if (job.hit_counters)
    {
        // camera rays of this launch that hit something: per workgroup through LDS, one global add per workgroup
        const uint32_t wave_hits = lanes_sum(hits);
        if ((threadIdx.x & 63u) == 0 && wave_hits)
            atomicAdd(&block_hits, wave_hits);
        __syncthreads();
        if (threadIdx.x == 0 && block_hits)
            atomicAdd(&job.hit_counters[blockIdx.x % kHitCounters], block_hits);
    }
    if (kCount)
    {
        atomicAdd(&counters->closest_rays, static_cast<unsigned long long>(rays));
        atomicAdd(&counters->node_tests, static_cast<unsigned long long>(ts.node_tests));
        atomicAdd(&counters->prim_tests, static_cast<unsigned long long>(ts.prim_tests));
        if (ts.wave_node_steps)
            atomicAdd(&counters->wave_node_steps, static_cast<unsigned long long>(ts.wave_node_steps));
        if (ts.wave_prim_steps)
            atomicAdd(&counters->wave_prim_steps, static_cast<unsigned long long>(ts.wave_prim_steps));
    }
}

template <bool kAnalytic, bool kSlivers, bool kVote, bool kCount>
hipError_t Launch(const DeviceScene &sc, const RenderJob &job, uint32_t *prehit, TraceCounters *counters, hipStream_t stream,
                  uint32_t n_cus)
{
    const size_t lds_bytes = size_t(sc.integrator.walk_depth) * kBlockSize * sizeof(uint32_t);
    int per_cu = 0;
    hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, primary_kernel<kAnalytic, kSlivers, kVote, kCount>,
                                                                  kBlockSize, lds_bytes);
    if (err != hipSuccess)
        return err;
    const unsigned long long total = static_cast<unsigned long long>(job.n_items) * sc.camera.spp;
    unsigned long long blocks = (total + kBlockSize - 1) / kBlockSize;
    const unsigned long long resident = static_cast<unsigned long long>(n_cus) * static_cast<unsigned>(per_cu < 1 ? 1 : per_cu);
    // a few grid-stride rounds per resident workgroup keep consecutive samples of a pixel in one wavefront and the tail short
    if (blocks > resident * 8)
        blocks = resident * 8;
    if (blocks == 0)
        return hipSuccess;
    hipLaunchKernelGGL((primary_kernel<kAnalytic, kSlivers, kVote, kCount>), dim3(static_cast<uint32_t>(blocks)), dim3(kBlockSize),
                       lds_bytes, stream, sc, job, prehit, counters);
    return hipGetLastError();
}

} // namespace

bool PrimaryPrepassSupports(const DeviceScene &sc, const RenderJob &job)
{
    // (an opacity mask draws a random number during the walk: the camera ray's hit is then part of the chain)
    return !job.reference_walk && !sc.integrator.has_masks && sc.integrator.n_walk_nodes != 0;
}

hipError_t LaunchPrimaryPrepass(const DeviceScene &sc, const RenderJob &job, uint32_t *prehit, TraceCounters *counters,
                                hipStream_t stream, uint32_t n_cus)
{
    const bool analytic = (sc.features & kFeatAnalytic) != 0, slivers = sc.integrator.walk_sliver_reach > 0.0f;
    const bool count = counters != nullptr;
    // always the vote-scheduled walk (it reads whole 64-byte records: the geometry comes from HBM / L1 here, also for the
    // scenes the render kernels stage in LDS; with both thresholds 0 it is the plain ordered walk plus one ballot per step)
    if (analytic)
        return count ? Launch<true, true, true, true>(sc, job, prehit, counters, stream, n_cus)
                     : Launch<true, true, true, false>(sc, job, prehit, counters, stream, n_cus);
    if (slivers)
        return count ? Launch<false, true, true, true>(sc, job, prehit, counters, stream, n_cus)
                     : Launch<false, true, true, false>(sc, job, prehit, counters, stream, n_cus);
    return count ? Launch<false, false, true, true>(sc, job, prehit, counters, stream, n_cus)
                 : Launch<false, false, true, false>(sc, job, prehit, counters, stream, n_cus);
}

} // namespace mcpt
