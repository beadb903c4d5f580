// Render kernels of the LOW-DISCREPANCY build (hip/lowdisc_units.h): the dispatcher of throughput mode 2 and the general
// instantiations — full feature set on the reference-order walk (opacity masks), on the ordered walk of LDS-resident scenes,
// and on the vote-scheduled walk with the sliver rules.
#define MCPT_UNIT_LOWDISC
#include "lowdisc_units.h"

namespace mcpt
{

template hipError_t Launch<kAll | kLD, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kAll | kO | kLD, false, true>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kAll | kV | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);

// The scene classes of LaunchRender (hip/render_kernel.hip), for the instantiations that exist in this build.
hipError_t LaunchRenderLowDiscrepancy(const DeviceScene &sc, const RenderJob &job, float *out, hipStream_t stream, uint32_t n_cus,
                                      const char **variant)
{
    if (job.independent_samples != 2u)
        return hipErrorInvalidValue;
    const uint32_t f = sc.features;
    if (job.reference_walk || sc.integrator.has_masks)
    {
        *variant = "all, reference walk, Sobol points";
        return Launch<kAll | kLD, false, false>(sc, job, out, nullptr, stream, n_cus);
    }
    const bool slivers = sc.integrator.walk_sliver_reach > 0.0f;
    const bool lds = StagedBytes(sc, true) <= kLdsGeometryBytes;
    if (job.pool_walk >= 1 && !lds && PoolBigSupports(sc))
    {
        // scenes outside LDS: the pool walk with 32-bit items (dragon/scene.xml, matpreview, the reference's box scene)
        if ((f & ~kSurface) != 0)
        {
            *variant = slivers ? "all+slivers+pool-walk, Sobol points" : "all+pool-walk, Sobol points";
            return slivers ? Launch<kAll | kPB | kS | kLD, false, false>(sc, job, out, nullptr, stream, n_cus)
                           : Launch<kAll | kPB | kLD, false, false>(sc, job, out, nullptr, stream, n_cus);
        }
        if ((f & ~kFeatEmitters) == 0)
        {
            *variant = slivers ? "diffuse-emitters+slivers+pool-walk, Sobol points" : "diffuse-emitters+pool-walk, Sobol points";
            return slivers ? Launch<kFeatEmitters | kPB | kS | kLD, false, false>(sc, job, out, nullptr, stream, n_cus)
                           : Launch<kFeatEmitters | kPB | kLD, false, false>(sc, job, out, nullptr, stream, n_cus);
        }
        *variant = slivers ? "surface-materials+slivers+pool-walk, Sobol points" : "surface-materials+pool-walk, Sobol points";
        return slivers ? Launch<kSurface | kPB | kS | kLD, false, false>(sc, job, out, nullptr, stream, n_cus)
                       : Launch<kSurface | kPB | kLD, false, false>(sc, job, out, nullptr, stream, n_cus);
    }
    if (lds && !slivers)
    {
        const bool pool = job.pool_walk != 0 && sc.integrator.n_pool_nodes != 0 && sc.integrator.n_pool_nodes <= kPoolMaxRef + 1u &&
                          sc.integrator.n_prims <= kPoolMaxRef + 1u && sc.integrator.pool_depth <= kPoolMaxDepth &&
                          StagedBytes(sc, true, true) <= kLdsGeometryBytes;
        if (pool && f == 0)
        {
            *variant = "diffuse-area+lds+pool-walk, Sobol points";
            return Launch<kP | kLD, false, true>(sc, job, out, nullptr, stream, n_cus);
        }
        if (pool && (f & ~kFeatEmitters) == 0)
        {
            *variant = "diffuse-emitters+lds+pool-walk, Sobol points";
            return Launch<kFeatEmitters | kP | kLD, false, true>(sc, job, out, nullptr, stream, n_cus);
        }
        if ((f & ~kVolumeLean) == 0)
        {
            *variant = "volume-quadrics-microfacet+lds, Sobol points";
            return Launch<kVolumeLean | kO | kLD, false, true>(sc, job, out, nullptr, stream, n_cus);
        }
        *variant = "all+lds, Sobol points";
        return Launch<kAll | kO | kLD, false, true>(sc, job, out, nullptr, stream, n_cus);
    }
    *variant = "all+slivers, Sobol points";
    return Launch<kAll | kV | kS | kLD, false, false>(sc, job, out, nullptr, stream, n_cus);
}

} // namespace mcpt
