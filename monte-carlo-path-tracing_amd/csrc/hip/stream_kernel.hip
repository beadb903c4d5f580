// Dispatcher of the stream kernel + the instantiations of unit 0 (stream_variants.inc lists them all).
#define MCPT_STREAM_UNIT 0
#include "stream_units.h"

#include <cstdlib>

namespace mcpt
{

namespace
{

struct StreamVariant
{
    uint32_t features, shadow;
    bool counted, small, regs;
    const char *name;
    hipError_t (*plan)(const DeviceScene &, const RenderJob &, uint32_t, StreamLaunch &);
    hipError_t (*launch)(const DeviceScene &, const RenderJob &, float *, TraceCounters *, hipStream_t, uint32_t *,
                         const StreamLaunch &);
};

#define X(index, features, S, counted, small, hot, regs, unit, name) \
    {features, S, counted, small, regs, name, &PlanStream<(features), S, counted, small, hot, regs>, \
     &LaunchStream<(features), S, counted, small, hot, regs>},
const StreamVariant kVariants[] = {
#include "stream_variants.inc"
};
#undef X
constexpr uint32_t kVariantCount = sizeof(kVariants) / sizeof(kVariants[0]);

} // namespace

bool StreamSupports(const DeviceScene &sc, const RenderJob &job)
{
    const uint32_t shadows = sc.integrator.n_emitters + (sc.integrator.n_area_lights ? 1u : 0u);
    return !job.reference_walk && !sc.integrator.has_masks && sc.integrator.n_walk_nodes != 0 && shadows <= kStreamMaxShadow;
}

bool StreamPrefersLanes(const DeviceScene &sc) { return StagedBytes(sc, true) <= kLdsGeometryBytes; }

hipError_t PlanRenderStream(const DeviceScene &sc, const RenderJob &job, bool counted, uint32_t n_cus, StreamLaunch *cfg,
                            const char **name)
{
    const bool regs = !cfg->slots_in_memory;
    if (!StreamSupports(sc, job))
        return hipErrorNotSupported;
    const uint32_t shadows = sc.integrator.n_emitters + (sc.integrator.n_area_lights ? 1u : 0u);
    const uint32_t need_s = shadows <= 1 ? 1u : 2u;
    const bool slivers = sc.integrator.walk_sliver_reach > 0.0f;
    const uint32_t f = sc.features & (kAll);
    const bool small = StagedBytes(sc, true) <= kLdsGeometryBytes;
    // the leanest instantiation that covers the scene: fewest feature bits, then fewest shadow records — at the register
    // budget asked for (StreamLaunch::waves; only the surface-materials mesh instantiations exist at 3 and 2: otherwise 4)
    static const bool full_bsdf_set = std::getenv("MCPT_STREAM_FULL_BSDF_SET") != nullptr; // (measurements: never the instantiations without transmissive BSDFs)
    int pick = -1;
    for (uint32_t want_waves = (cfg->waves == 2 || cfg->waves == 3) ? cfg->waves : 4u; pick < 0; want_waves = 4u)
    {
    for (uint32_t v = 0; v < kVariantCount; ++v)
    {
        const StreamVariant &k = kVariants[v];
        const uint32_t kf = k.features & kAll;
        if (k.counted != counted || k.small != small || k.regs != regs || (f & ~kf) != 0 || k.shadow < need_s)
            continue;
        if (slivers && !(k.features & kFeatSlivers))
            continue;
        if ((k.features & kFeatNoTransmission) && (sc.integrator.has_transmission || full_bsdf_set))
            continue;
        if ((k.features & kFeatDielectricOnly) && (sc.integrator.has_reflectors || full_bsdf_set))
            continue;
        if ((k.features & kFeatConductorOnly) && (sc.integrator.has_non_conductor || full_bsdf_set))
            continue;
        if (((k.features & kFeatWaves2) ? 2u : (k.features & kFeatWaves3) ? 3u : 4u) != want_waves)
            continue;
        if (pick < 0)
        {
            pick = static_cast<int>(v);
            continue;
        }
        const StreamVariant &b = kVariants[pick];
        const int cost_k = __builtin_popcount(k.features & ~(kFeatWaves2 | kFeatWaves3 | kFeatNoTransmission | kFeatDielectricOnly | kFeatConductorOnly)) * 4 + static_cast<int>(k.shadow) - ((k.features & kFeatNoTransmission) ? 1 : 0) - ((k.features & (kFeatDielectricOnly | kFeatConductorOnly)) ? 2 : 0);
        const int cost_b = __builtin_popcount(b.features & ~(kFeatWaves2 | kFeatWaves3 | kFeatNoTransmission | kFeatDielectricOnly | kFeatConductorOnly)) * 4 + static_cast<int>(b.shadow) - ((b.features & kFeatNoTransmission) ? 1 : 0) - ((b.features & (kFeatDielectricOnly | kFeatConductorOnly)) ? 2 : 0);
        if (cost_k < cost_b)
            pick = static_cast<int>(v);
    }
    if (want_waves == 4u)
        break;
    }
    if (pick < 0)
        return hipErrorNotSupported;
    const StreamVariant &k = kVariants[pick];
    cfg->variant = static_cast<uint32_t>(pick);
    *name = k.name;
    // Launch shape.  Small scenes: 2 slots per lane keep two workgroups on a CU (LDS: traversal data + stacks +
    // hot fields), which measured best; meshes: 2 slots per lane, everything but the stacks and the ray list in
    // cached global memory.  A wavefront refills when 24 of its lanes are free.
    if (cfg->slots_in_memory)
        cfg->wave_local = 0;
    if (cfg->slots == 0)
        cfg->slots = 2 * kBlockSize;
    cfg->slots = ((cfg->slots + kBlockSize - 1) / kBlockSize) * kBlockSize;
    if (cfg->refill_at == 0)
        cfg->refill_at = 24; // (swept 8 .. 64 with the work counter and the pre-pass: flat from 16 to 32, -12 % at 64, -9 % at 8 on dragon)
    return k.plan(sc, job, n_cus, *cfg);
}

hipError_t LaunchRenderStream(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters,
                              hipStream_t stream, uint32_t *scratch, const StreamLaunch &cfg)
{
    if (cfg.variant >= kVariantCount)
        return hipErrorInvalidValue;
    return kVariants[cfg.variant].launch(sc, job, out, counters, stream, scratch, cfg);
}

} // namespace mcpt
