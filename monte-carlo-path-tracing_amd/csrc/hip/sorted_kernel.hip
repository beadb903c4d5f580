// The class-sorted kernel's entry points and launcher; the body: hip/sorted_body.h.
#include "sorted_body.h"

namespace mcpt
{

// (two forms, overloads of one name: the records by value, or behind a pointer to device memory — render_kernel.h LaunchRecords,
//  render_kernel_impl.h records_behind_pointer; every class-sorted instantiation takes the pointer: volumetric-caustic 649 -> 637 ms)
template <uint32_t kFeatures, bool kLdsGeometry>
__global__ void __launch_bounds__(kSortLanes, sorted_waves(kFeatures))
sorted_kernel(const DeviceScene sc_in, const RenderJob job, float *__restrict__ out)
{
    sorted_body<kFeatures, kLdsGeometry>(sc_in, job, out);
}
template <uint32_t kFeatures, bool kLdsGeometry>
__global__ void __launch_bounds__(kSortLanes, sorted_waves(kFeatures))
sorted_kernel(LaunchRecordsPtr records, float *__restrict__ out)
{
    sorted_body<kFeatures, kLdsGeometry>(MCPT_RECORDS_SCENE(records), MCPT_RECORDS_JOB(records), out);
}
template <uint32_t kFeatures>
constexpr bool sorted_behind_pointer()
{
#if defined(MCPT_SCENE_POINTER)
    return MCPT_SCENE_POINTER != 0;
#else
    return true;
#endif
}

template <uint32_t kFeatures, bool kLdsGeometry = true>
static hipError_t LaunchSorted(const DeviceScene &sc, const RenderJob &job, float *out, hipStream_t stream, uint32_t max_blocks)
{
    constexpr uint32_t kBlockSize = kSortLanes;
    const size_t lds_bytes = SortedLdsBytes<kFeatures, kLdsGeometry>(sc);
    int per_cu = 0;
    constexpr bool kSortedByPointer = sorted_behind_pointer<kFeatures>();
    using ByValue = void (*)(const DeviceScene, const RenderJob, float *);
    using ByPointer = void (*)(LaunchRecordsPtr, float *);
    const typename std::conditional<kSortedByPointer, ByPointer, ByValue>::type kernel = sorted_kernel<kFeatures, kLdsGeometry>;
    hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlockSize, lds_bytes);
    if (err != hipSuccess)
        return err;
    if (per_cu < 1)
        return hipErrorOutOfMemory;
    const uint32_t n_work = job.n_items * (job.sample_split ? job.sample_split : 1u);
    const uint32_t resident = max_blocks * static_cast<uint32_t>(per_cu);
    RenderJob j = job;
    j.lane_spread = 1;
    if (j.scatter == kScatterAuto)
        j.scatter = 0;
    NoteTransposed(j.scatter != 0);
    uint64_t blocks = (uint64_t(n_work) + kBlockSize - 1) / kBlockSize;
    if (blocks > resident)
        blocks = resident;
    if (blocks == 0)
        return hipSuccess;
    const dim3 grid(static_cast<uint32_t>(blocks)), block(kBlockSize);
    if constexpr (kSortedByPointer)
    {
        err = StageLaunchRecords(sc, j, stream);
        if (err != hipSuccess)
            return err;
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, (LaunchRecordsPtr)(j.launch_records), out);
        return hipGetLastError();
    }
    else
    {
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, sc, j, out);
        return hipGetLastError();
    }
}

// The class-sorted kernel for the job's scene, or hipErrorNotSupported when the scene is not one of its classes (the
// caller then takes render_kernel): full-feature scenes with the traversal data in LDS, ordered walk, no counters.
hipError_t LaunchRenderSorted(const DeviceScene &sc, const RenderJob &job, float *out, hipStream_t stream, uint32_t n_cus,
                              const char **variant)
{
    const uint32_t f = sc.features;
    constexpr uint32_t kG = kFeatGroup128;
    if (job.reference_walk || sc.integrator.has_masks || sc.integrator.walk_sliver_reach > 0.0f || (f & (kFeatVolPath | kFeatAnalytic | kFeatMicrofacet)) == 0)
        return hipErrorNotSupported;
    if (StagedBytes(sc, true) > kLdsGeometryBytes)
    {
        // OUTSIDE LDS (round 6): surface-material meshes — matpreview, BASELINE's "BSDF-sort path" — on the pool walk with 32-bit items.
        // Only on request (RenderJob::sort_classes 2 = mcpt_renderer_set_class_sort(r, 1)): measured slower than the unsorted kernels with
        // lanes per path by tile cost (EXPERIMENTS R6-4).
        if (job.sort_classes < 2 || (f & ~kSurface) != 0 || !PoolBigSupports(sc))
            return hipErrorNotSupported;
        if (!sc.integrator.has_non_conductor)
        {
            *variant = "surface-materials (diffuse + conductor only)+pool-walk, class-sorted";
            return LaunchSorted<kSurface | kPBU | kFeatConductorOnly | kG, false>(sc, job, out, stream, n_cus);
        }
        if (!sc.integrator.has_reflectors)
        {
            *variant = "surface-materials (diffuse + dielectric only)+pool-walk, class-sorted";
            return LaunchSorted<kSurface | kPBU | kFeatDielectricOnly | kG, false>(sc, job, out, stream, n_cus);
        }
        *variant = "surface-materials+pool-walk, class-sorted";
        return LaunchSorted<kSurface | kPBU | kG, false>(sc, job, out, stream, n_cus);
    }
    // the wavefront-cooperative pool walk (pool_walk.h): its items hold node and slot indices in 10 bits
    const bool pool = job.pool_walk >= 2 && sc.integrator.n_pool_nodes != 0 && sc.integrator.n_pool_nodes <= kPoolMaxRef + 1u &&
                      sc.integrator.n_prims <= kPoolMaxRef + 1u && sc.integrator.pool_depth <= kPoolMaxDepth && StagedBytes(sc, true, true) <= kLdsGeometryBytes;
    if ((f & ~kVolumeLean) == 0)
    {
        *variant = pool ? "volume-quadrics-microfacet+lds+pool-walk, class-sorted" : "volume-quadrics-microfacet+lds, class-sorted";
        return pool ? LaunchSorted<kVolumeLean | kP | kG>(sc, job, out, stream, n_cus) : LaunchSorted<kVolumeLean | kO | kG>(sc, job, out, stream, n_cus);
    }
    *variant = pool ? "all+lds+pool-walk, class-sorted" : "all+lds, class-sorted";
    return pool ? LaunchSorted<kAll | kP | kG>(sc, job, out, stream, n_cus) : LaunchSorted<kAll | kO | kG>(sc, job, out, stream, n_cus);
}

} // namespace mcpt
