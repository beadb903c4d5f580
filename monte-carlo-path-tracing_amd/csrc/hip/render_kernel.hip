// The lean render-kernel instantiations, the unit kernels and the dispatcher; the kernel template
// itself is in render_kernel_impl.h.
#include <cstddef>
#include <cstdlib>

#include "render_kernel_impl.h"
#include "../host/measurement_env.hpp"

namespace mcpt
{

// ---- the records of a launch in device memory (render_kernel.h, LaunchRecords) ----------
// The argument segment of this kernel IS a LaunchRecords (the two records by value, each at its natural alignment): its 64 lanes copy it.
__global__ void stage_records_kernel(const DeviceScene, const RenderJob, uint32_t *__restrict__ dst)
{
    static_assert(offsetof(LaunchRecords, job) == (sizeof(DeviceScene) + alignof(RenderJob) - 1) / alignof(RenderJob) * alignof(RenderJob), "two kernel arguments lie like the struct's two members");
    const uint32_t __attribute__((address_space(4))) *src = (const uint32_t __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    for (uint32_t i = threadIdx.x; i < sizeof(LaunchRecords) / sizeof(uint32_t); i += blockDim.x)
        dst[i] = src[i];
}

hipError_t StageLaunchRecords(const DeviceScene &sc, const RenderJob &job, hipStream_t stream)
{
    if (!job.launch_records)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(stage_records_kernel, dim3(1), dim3(64), 0, stream, sc, job, reinterpret_cast<uint32_t *>(job.launch_records));
    return hipGetLastError();
}

// ---- unit kernels (diagnostics / parity tests): one query per lane ----------

// in: origin[3], dir[3] per ray.  out (19 floats per ray): valid, inside,
// instance, primitive-in-instance, t, uv[2], position, normal, tangent, bitangent.
template <bool kOrdered>
__global__ void intersect_kernel(const DeviceScene sc, uint32_t n, const float *__restrict__ rays,
                                 const uint32_t *__restrict__ seeds, float *__restrict__ out,
                                 uint32_t *__restrict__ seeds_out)
{
    extern __shared__ uint32_t lds_stack[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    Ray ray = make_ray(V3{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, V3{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
    uint32_t rng = seeds[i];
    HitRaw raw;
    TraceStats ts{0, 0, 0, 0};
    const bool hit = kOrdered ? walk_ordered<false, true, false>(sc, lds_stack + threadIdx.x, ray, raw, ts)
                              : walk_scene<false, true, true, false>(sc, ray, rng, raw, ts);
    float *o = out + 19 * static_cast<size_t>(i);
    for (int k = 0; k < 19; ++k)
        o[k] = 0.0f;
    o[2] = o[3] = -1.0f;
    o[4] = ray.t_max;
    seeds_out[i] = rng;
    if (!hit)
        return;
    const Surface s = make_surface<true, true>(sc, ray, raw);
    o[0] = 1.0f, o[1] = s.inside ? 1.0f : 0.0f, o[2] = static_cast<float>(s.inst);
    o[3] = static_cast<float>(raw.prim - sc.instances[s.inst].prim_base);
    o[5] = s.uv.u, o[6] = s.uv.v;
    const V3 v[4] = {s.position, s.normal, s.tangent, s.bitangent};
    for (int k = 0; k < 4; ++k)
        o[7 + 3 * k] = v[k].x, o[8 + 3 * k] = v[k].y, o[9 + 3 * k] = v[k].z;
}

// The same through the wavefront-cooperative pool walk (pool_walk.h) in its most general form: 32-bit items, the 4-wide
// exact hierarchy through the caches, quadrics, sliver rules.  Every lane of a wavefront makes the call (the ones beyond
// n only work on the others' rays).
__global__ void intersect_pool_kernel(const DeviceScene sc, uint32_t n, const float *__restrict__ rays, float *__restrict__ out)
{
    extern __shared__ uint32_t lds_pool[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = i < n;
    const uint32_t k = mine ? i : 0u;
    Ray ray = make_ray(V3{rays[6 * k], rays[6 * k + 1], rays[6 * k + 2]}, V3{rays[6 * k + 3], rays[6 * k + 4], rays[6 * k + 5]});
    HitRaw raw;
    raw.inst = raw.prim = 0, raw.a = raw.b = raw.c = 0.0f, raw.inside = false;
    TraceStats ts{0, 0, 0, 0};
    uint32_t *pool = lds_pool + (threadIdx.x >> 6) * pool_wave_words(true, true);
    const bool hit = sc.integrator.walk_sliver_reach > 0.0f ? walk_pool<false, true, false, true, true>(sc, pool, mine, ray, raw, ts)
                                                           : walk_pool<false, true, false, true, false>(sc, pool, mine, ray, raw, ts);
    if (!mine)
        return;
    float *o = out + 19 * static_cast<size_t>(i);
    for (int c = 0; c < 19; ++c)
        o[c] = 0.0f;
    o[2] = o[3] = -1.0f;
    o[4] = ray.t_max;
    if (!hit)
        return;
    const Surface s = make_surface<true, true>(sc, ray, raw);
    o[0] = 1.0f, o[1] = s.inside ? 1.0f : 0.0f, o[2] = static_cast<float>(s.inst);
    o[3] = static_cast<float>(raw.prim - sc.instances[s.inst].prim_base);
    o[5] = s.uv.u, o[6] = s.uv.v;
    const V3 v[4] = {s.position, s.normal, s.tangent, s.bitangent};
    for (int c = 0; c < 4; ++c)
        o[7 + 3 * c] = v[c].x, o[8 + 3 * c] = v[c].y, o[9 + 3 * c] = v[c].z;
}

// The steps of one pixel, one lane, reference-order walk (both walks give the same frame):
// per step 16 floats {ray origin[3], ray direction[3], throughput max, hit primitive (-1 = none),
// distance, shadow queries, last shadow result, LCG state after the step (bits), L[3] so far,
// depth}.  n_steps[0] = steps written.  The CPU build of the same code is tests/emu's
// mcpt_emu_debug_pixel: comparing the two shows where a device frame leaves the host's.
__global__ void trace_pixel_kernel(const DeviceScene sc, uint32_t pixel, uint32_t capacity, float *__restrict__ out,
                                   uint32_t *__restrict__ n_steps)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    PathState st;
    st.stack = nullptr;
    LaneCounters cnt{};
    start_pixel(st, pixel);
    uint32_t n = 0;
    while (!pixel_done(sc, st) && n < capacity)
    {
        if (!st.alive)
            start_sample(sc, st);
        path_step<Config<kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet>>(sc, st, &cnt);
        float *o = out + 16 * static_cast<size_t>(n++);
        for (int k = 0; k < 6; ++k)
            o[k] = cnt.last_closest_ray[k];
        o[6] = max_component(st.throughput);
        o[7] = cnt.last_hit_prim == kNone ? -1.0f : static_cast<float>(cnt.last_hit_prim);
        o[8] = cnt.last_hit_t, o[9] = static_cast<float>(cnt.last_shadow_count);
        o[10] = static_cast<float>(cnt.last_shadow_hit);
        o[11] = __uint_as_float(st.rng);
        o[12] = st.L.x, o[13] = st.L.y, o[14] = st.L.z, o[15] = static_cast<float>(st.depth);
    }
    n_steps[0] = n;
}

hipError_t LaunchTracePixel(const DeviceScene &sc, uint32_t pixel, uint32_t capacity, float *out, uint32_t *n_steps,
                            hipStream_t stream)
{
    hipLaunchKernelGGL(trace_pixel_kernel, dim3(1), dim3(64), 0, stream, sc, pixel, capacity, out, n_steps);
    return hipGetLastError();
}

// in (18 floats per query): wo, wi, normal, tangent, bitangent, uv, inside.
// out (8 floats): valid, pdf, attenuation[3], wi[3].  mode 0 = evaluate, 1 = sample.
__global__ void bsdf_kernel(const DeviceScene sc, uint32_t n, uint32_t id_bsdf, int mode,
                            const float *__restrict__ recs, const uint32_t *__restrict__ seeds,
                            float *__restrict__ out, uint32_t *__restrict__ seeds_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float *r = recs + 18 * static_cast<size_t>(i);
    BsdfQuery q;
    q.valid = false, q.pdf = 0, q.attenuation = V3{0, 0, 0};
    q.wo = V3{r[0], r[1], r[2]}, q.wi = V3{r[3], r[4], r[5]}, q.normal = V3{r[6], r[7], r[8]};
    q.tangent = V3{r[9], r[10], r[11]}, q.bitangent = V3{r[12], r[13], r[14]};
    q.uv = V2{r[15], r[16]}, q.inside = r[17] != 0.0f;
    uint32_t rng = seeds[i];
    const ShadeTables T = ShadeTables{sc.textures, sc.texels, sc.lut_brdf, sc.lut_albedo, false};
    if (mode == 0)
        bsdf_eval<true>(T, sc.bsdfs[id_bsdf], q);
    else
        bsdf_sample<true>(T, sc.bsdfs[id_bsdf], rng, q);
    float *o = out + 8 * static_cast<size_t>(i);
    o[0] = q.valid ? 1.0f : 0.0f, o[1] = q.pdf;
    o[2] = q.attenuation.x, o[3] = q.attenuation.y, o[4] = q.attenuation.z;
    o[5] = q.wi.x, o[6] = q.wi.y, o[7] = q.wi.z;
    seeds_out[i] = rng;
}

hipError_t LaunchIntersect(const DeviceScene &sc, uint32_t n, const float *rays, const uint32_t *seeds, float *out,
                           uint32_t *seeds_out, bool reference_walk, hipStream_t stream, bool pool_walk)
{
    if (n == 0)
        return hipSuccess;
    if (pool_walk && !reference_walk && !sc.integrator.has_masks)
    {
        if (sc.integrator.n_pool_nodes == 0 || sc.integrator.n_pool_nodes > kPoolMaxRefBig || sc.integrator.pool_depth > kPoolMaxDepth ||
            (MCPT_POOL_QUANT != 0 && (sc.integrator.n_wide_nodes == 0 || sc.integrator.n_wide_nodes > kPoolMaxRefBig)))
            return hipErrorNotSupported;
        hipLaunchKernelGGL(intersect_pool_kernel, dim3((n + 255) / 256), dim3(256), 4 * pool_wave_words(true, true) * sizeof(uint32_t), stream, sc, n,
                           rays, out);
        (void)hipMemcpyAsync(seeds_out, seeds, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream); // (the ordered walks draw nothing)
        return hipGetLastError();
    }
    if (reference_walk || sc.integrator.has_masks)
        hipLaunchKernelGGL(intersect_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, rays, seeds,
                           out, seeds_out);
    else
        hipLaunchKernelGGL(intersect_kernel<true>, dim3((n + 255) / 256), dim3(256),
                           sc.integrator.walk_depth * 256 * sizeof(uint32_t), stream, sc, n, rays, seeds, out, seeds_out);
    return hipGetLastError();
}

hipError_t LaunchBsdf(const DeviceScene &sc, uint32_t n, uint32_t id_bsdf, int mode, const float *recs,
                      const uint32_t *seeds, float *out, uint32_t *seeds_out, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(bsdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, id_bsdf, mode, recs, seeds, out,
                       seeds_out);
    return hipGetLastError();
}

// Picks the leanest instantiation that covers the scene's feature bits.  The
// ordered walk is the default; opacity masks (or job.reference_walk) select the
// reference-order walk, which exists only in the full instantiation.  The two
// LDS-resident instantiations (tiny scenes) use the plain ordered walk, all others
// the vote-scheduled one (its threshold comes from the commit: 0 for small scenes).
namespace
{
thread_local bool g_last_transposed = false;
}
void NoteTransposed(bool transposed) { g_last_transposed = transposed; }
bool LastLaunchTransposed() { return g_last_transposed; }

// (measurements: MCPT_POOL_SUBSETS=0 keeps the full BSDF set in the pool-walk kernels of scenes that need one model only)
static bool PoolSubsets()
{
    static const bool on = []
    {
        const char *e = mcpt::MeasurementEnv("MCPT_POOL_SUBSETS");
        return !e || std::atoi(e) != 0;
    }();
    return on;
}

// The lean LDS-resident pool-walk kernels (cornell's class) exist in two forms: two queries per vertex (kP) and merged queries
// (kPM: one ray record more per lane — 10.9 instead of 7.8 KB of LDS per wavefront, three workgroups per CU instead of four).
// RenderJob::pool_walk bit 2 asks for the merged form (mcpt_renderer_set_pool_walk(r, 2); EXPERIMENTS R6-1 / R6-2: 27 % slower on a
// full frame, which is bound by VALU issue at four wavefronts per SIMD — what merged queries shorten is a path's chain).

// (the references an item of the production pool walks can hold: one bit less with merged queries, pool_walk.h)
constexpr uint32_t kLdsPoolLimit = kPoolMaxRef;
constexpr uint32_t kBigPoolLimit = kPoolMaxRefBigDual; // (the merged instantiations' 25 bits, for every kernel of the class)

// Can the lane-owns-a-path kernel run this scene (outside LDS) with the pool walk?  No opacity masks (they draw random numbers
// during a walk: the visiting order is part of the image), the 4-wide hierarchy within the items' 26 bits and the lists' head room.
bool PoolBigSupports(const DeviceScene &sc)
{
    // (MCPT_POOL_QUANT: the node items name records of wide_nodes — the same collapse, plus its padding records)
    return !sc.integrator.has_masks && sc.integrator.n_pool_nodes != 0 && sc.integrator.n_pool_nodes <= kBigPoolLimit &&
           (MCPT_POOL_QUANT == 0 || (sc.integrator.n_wide_nodes != 0 && sc.integrator.n_wide_nodes <= kBigPoolLimit)) &&
           sc.integrator.n_prims <= kBigPoolLimit && sc.integrator.pool_depth <= kPoolMaxDepth;
}

// (LaunchRender's own conditions for Launch<kFeatEmitters | kPB [| kS]>, below)
bool TailSpreadRuns(const DeviceScene &sc, const RenderJob &job)
{
    return MCPT_TAIL_SPREAD != 0 && !job.reference_walk && !sc.integrator.has_masks && job.pool_walk >= 1 && StagedBytes(sc, true) > kLdsGeometryBytes && PoolBigSupports(sc) &&
           (sc.features & ~uint32_t(MCPT_TAIL_SPREAD == 2 ? kSurface : kFeatEmitters)) == 0;
}

hipError_t LaunchRender(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters,
                        hipStream_t stream, uint32_t n_cus, const char **variant)
{
    const uint32_t f = sc.features;
    const bool ordered = !job.reference_walk && !sc.integrator.has_masks;
#if defined(MCPT_EXPERIMENT_LEAN_ONLY)
    // developer builds for kernel experiments: only the cornell instantiation (fast to compile)
    *variant = "diffuse-area+lds (experiment build)";
    if (counters != nullptr || !ordered || f != 0 || StagedBytes(sc, true) > kLdsGeometryBytes)
        return hipErrorNotSupported;
    return job.pool_walk ? Launch<kP, false, true>(sc, job, out, nullptr, stream, n_cus) : Launch<kO, false, true>(sc, job, out, nullptr, stream, n_cus);
#endif
    const bool slivers = sc.integrator.walk_sliver_reach > 0.0f;
    // (Round 5's advisor: "follows" means the KIND of ray query — pool walk or per-lane walk.  The counting instantiations carry the
    //  volume-path code, so they are never merged (Config::kPoolDual needs !kVolPath): they walk two queries per vertex where the
    //  production kernels outside LDS walk one merged query — the same rays, box and primitive tests, twice the query rounds — and
    //  bench.py says so next to the counts.)
    // The counting instantiation follows the PRODUCTION launch's ray query (round 4's advisor: it used to count a pool walk for every
    // scene the big form supports, also where the timed kernel walks per lane — volumetric-caustic's class-sorted kernel): the pool
    // walk's per-item counts (4 box tests per node item) where the uncounted launch below uses the pool walk — its LDS form
    // (exact 128-byte records) for the lean LDS-resident scenes, the quantised form outside LDS —, the per-lane walk's otherwise.
    const bool lds_scene = StagedBytes(sc, true) <= kLdsGeometryBytes;
    const bool lds_pool = lds_scene && !slivers && job.pool_walk != 0 && sc.integrator.n_pool_nodes != 0 && sc.integrator.n_pool_nodes <= kLdsPoolLimit + 1u &&
                          sc.integrator.n_prims <= kLdsPoolLimit + 1u && sc.integrator.pool_depth <= kPoolMaxDepth && StagedBytes(sc, true, true) <= kLdsGeometryBytes &&
                          ((f & ~kFeatEmitters) == 0 || job.pool_walk >= 2);
    if (counters != nullptr && ordered && lds_pool)
    {
        *variant = "all+count+lds+pool-walk";
        return Launch<kAll | kP, true, true>(sc, job, out, counters, stream, n_cus);
    }
    if (counters != nullptr && ordered && !lds_scene && job.pool_walk >= 1 && PoolBigSupports(sc))
    {
        *variant = "all+count+pool-walk";
        return Launch<kAll | kPB | kS, true, false>(sc, job, out, counters, stream, n_cus);
    }
    if (counters != nullptr)
    {
        *variant = ordered ? "all+count" : "all+count, reference walk";
        return ordered ? Launch<kAll | kV | kS, true>(sc, job, out, counters, stream, n_cus)
                       : Launch<kAll, true>(sc, job, out, counters, stream, n_cus);
    }
    if (!ordered)
    {
        *variant = "all, reference walk";
        return Launch<kAll, false>(sc, job, out, nullptr, stream, n_cus);
    }
    // scenes outside LDS of the class the pool walk runs on (32-bit items, the 4-wide exact hierarchy through the caches): the
    // surface-materials instantiation with it, whatever subset of its features the scene uses
    if (job.pool_walk >= 1 && StagedBytes(sc, true) > kLdsGeometryBytes && PoolBigSupports(sc))
    {
        if ((f & ~kSurface) != 0)
        {
            *variant = slivers ? "all+slivers+pool-walk" : "all+pool-walk";
            return slivers ? Launch<kAll | kPB | kS, false>(sc, job, out, nullptr, stream, n_cus) : Launch<kAll | kPB, false>(sc, job, out, nullptr, stream, n_cus);
        }
        if ((f & ~kFeatEmitters) == 0)
        {
            // diffuse surfaces only (dragon/scene.xml): the lean instantiations
            *variant = slivers ? "diffuse-emitters+slivers+pool-walk" : "diffuse-emitters+pool-walk";
            return slivers ? Launch<kFeatEmitters | kPB | kS, false>(sc, job, out, nullptr, stream, n_cus) : Launch<kFeatEmitters | kPB, false>(sc, job, out, nullptr, stream, n_cus);
        }
        if (!slivers && PoolSubsets() && !sc.integrator.has_non_conductor)
        {
            *variant = "surface-materials (diffuse + conductor only)+pool-walk";
            return Launch<kSurface | kPBU | kFeatConductorOnly, false>(sc, job, out, nullptr, stream, n_cus);
        }
        if (!slivers && PoolSubsets() && !sc.integrator.has_reflectors)
        {
            *variant = "surface-materials (diffuse + dielectric only)+pool-walk";
            return Launch<kSurface | kPBU | kFeatDielectricOnly, false>(sc, job, out, nullptr, stream, n_cus);
        }
        *variant = slivers ? "surface-materials+slivers+pool-walk" : "surface-materials+pool-walk";
        return slivers ? Launch<kSurface | kPB | kS, false>(sc, job, out, nullptr, stream, n_cus) : Launch<kSurface | kPB, false>(sc, job, out, nullptr, stream, n_cus);
    }
    if (slivers)
    {
        // sliver triangles: the instantiations whose primitive test replays the reference on them
        if ((f & ~kSurface) == 0)
        {
            *variant = "surface-materials+slivers";
            return Launch<kSurface | kV | kS, false>(sc, job, out, nullptr, stream, n_cus);
        }
        *variant = "all+slivers";
        return Launch<kAll | kV | kS, false>(sc, job, out, nullptr, stream, n_cus);
    }
    const bool lds = StagedBytes(sc, true) <= kLdsGeometryBytes;
    // the wavefront-cooperative pool walk (pool_walk.h): its items hold node and slot indices in 10 bits
#if defined(MCPT_FORCE_LEAN_MERGED) // (experiment builds: tools/experiments/bisect_lean_merge.sh)
    const bool lean_merged = true;
#else
    const bool lean_merged = (job.pool_walk & 4u) != 0;
#endif
    const uint32_t lean_limit = lean_merged ? kPoolMaxRefDual : kPoolMaxRef; // (an item's reference bits: pool_walk.h)
    const bool pool = lds && job.pool_walk != 0 && sc.integrator.n_pool_nodes != 0 && sc.integrator.n_pool_nodes <= lean_limit + 1u &&
                      sc.integrator.n_prims <= lean_limit + 1u && sc.integrator.pool_depth <= kPoolMaxDepth &&
                      StagedBytes(sc, true, true) <= kLdsGeometryBytes;
    if (f == 0)
    {
        *variant = pool ? (lean_merged ? "diffuse-area+lds+pool-walk, merged queries" : "diffuse-area+lds+pool-walk") : lds ? "diffuse-area+lds" : "diffuse-area";
        return pool  ? (lean_merged ? Launch<kPM, false, true>(sc, job, out, nullptr, stream, n_cus) : Launch<kP, false, true>(sc, job, out, nullptr, stream, n_cus))
               : lds ? Launch<kO, false, true>(sc, job, out, nullptr, stream, n_cus)
                     : Launch<kV, false>(sc, job, out, nullptr, stream, n_cus);
    }
    if ((f & ~kFeatEmitters) == 0)
    {
        *variant = pool ? (lean_merged ? "diffuse-emitters+lds+pool-walk, merged queries" : "diffuse-emitters+lds+pool-walk") : lds ? "diffuse-emitters+lds" : "diffuse-emitters";
        return pool  ? (lean_merged ? Launch<kFeatEmitters | kPM, false, true>(sc, job, out, nullptr, stream, n_cus) : Launch<kFeatEmitters | kP, false, true>(sc, job, out, nullptr, stream, n_cus))
               : lds ? Launch<kFeatEmitters | kO, false, true>(sc, job, out, nullptr, stream, n_cus)
                     : Launch<kFeatEmitters | kV, false>(sc, job, out, nullptr, stream, n_cus);
    }
    if ((f & ~kSurface) == 0)
    {
        *variant = "surface-materials";
        return Launch<kSurface | kV, false>(sc, job, out, nullptr, stream, n_cus);
    }
    if (lds && (f & ~kVolumeLean) == 0)
    {
        // no emitter records, constant textures only: the instantiation without that code keeps its state in registers
        // (volumetric-caustic: 152 -> 6 spilled VGPRs at the same budget)
        *variant = "volume-quadrics-microfacet+lds";
        return Launch<kVolumeLean | kO, false, true>(sc, job, out, nullptr, stream, n_cus);
    }
    *variant = lds ? "all+lds" : "all";
    return lds ? Launch<kAll | kO, false, true>(sc, job, out, nullptr, stream, n_cus)
               : Launch<kAll | kV, false>(sc, job, out, nullptr, stream, n_cus);
}

// Tiles of one rank's packed block (64 pixels x 3 floats per tile, tile t = first + k * stride) -> their
// pixels of the frame.  One lane per float: both sides are touched in runs of 24 consecutive floats (a
// tile row), the packed side fully coalesced.
__global__ void __launch_bounds__(kBlockSize) unpack_tiles_kernel(const float *__restrict__ packed, float *__restrict__ frame,
                                                                  uint32_t tile_first, uint32_t tile_stride, uint32_t n_tiles,
                                                                  uint32_t tiles_x, uint32_t width, uint32_t height)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; // float index in the packed block
    if (i >= n_tiles * 192u)
        return;
    const uint32_t k = i / 192u, within = i % 192u, r = within / 3u, c = within % 3u;
    const uint32_t t = tile_first + k * tile_stride;
    const uint32_t x = (t % tiles_x) * 8u + (r & 7u), y = (t / tiles_x) * 8u + (r >> 3);
    if (x < width && y < height)
        frame[3 * (static_cast<size_t>(y) * width + x) + c] = packed[i];
}

hipError_t LaunchUnpackTiles(const float *packed, float *frame, uint32_t tile_first, uint32_t tile_stride, uint32_t n_tiles,
                             uint32_t tiles_x, uint32_t width, uint32_t height, hipStream_t stream)
{
    if (n_tiles == 0)
        return hipSuccess;
    const uint32_t n = n_tiles * 192u;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3((n + kBlockSize - 1) / kBlockSize), dim3(kBlockSize), 0, stream, packed, frame,
                       tile_first, tile_stride, n_tiles, tiles_x, width, height);
    return hipGetLastError();
}

// Independent-sample RNG mode with the samples of a pixel spread over `split` lanes: the lanes' unnormalised
// sums, added in lane order (deterministic), times 1 / spp.
__global__ void __launch_bounds__(kBlockSize) reduce_sample_planes_kernel(const float *__restrict__ planes, float *__restrict__ frame,
                                                                          uint32_t n_floats, uint32_t split, size_t plane_floats,
                                                                          float spp_inv)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats)
        return;
    float sum = planes[i];
    for (uint32_t k = 1; k < split; ++k)
        sum += planes[k * plane_floats + i];
    frame[i] = sum * spp_inv;
}

hipError_t LaunchReduceSamplePlanes(const float *planes, float *frame, uint32_t n_pixels, uint32_t split, uint32_t plane_stride,
                                    float spp_inv, hipStream_t stream)
{
    const uint32_t n = n_pixels * 3u;
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(reduce_sample_planes_kernel, dim3((n + kBlockSize - 1) / kBlockSize), dim3(kBlockSize), 0, stream, planes,
                       frame, n, split, static_cast<size_t>(plane_stride) * 3u, spp_inv);
    return hipGetLastError();
}

} // namespace mcpt
