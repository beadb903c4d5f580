// The streaming path-tracing state machine executed by every GPU lane.
//
// One lane owns one pixel at a time and walks through its `spp` samples in
// order, because the reference threads ONE LCG state through all samples of a
// pixel (renderer.cpp:62-85): the number of draws per sample is data
// dependent, so samples of a pixel cannot be computed out of order without
// changing the image.  Parallelism therefore comes from pixels.
//
// Instead of the reference's recursive shape (Shade -> direct light -> any-hit
// inside a per-depth loop), a lane runs a flat loop of STEPS.  Every step is
//      [regenerate]  camera ray for the next sample / next pixel, if needed
//      extend        closest-hit walk of the current ray
//      resolve       miss / emitter / back face / medium event bookkeeping
//      roulette      the reference's loop condition (path.cpp:57-59)
//      connect       next-event estimation: one shadow walk per emitter + one
//                    for the sampled area light
//      scatter       BSDF or phase-function sample -> next ray
// so all 64 lanes of a wavefront are in the same stage at the same time
// (extend with extend, connect with connect) no matter how long their
// individual paths are; a lane whose path ends simply regenerates at the top
// of the next step while its neighbours extend.  The arithmetic and the order
// of random draws inside each stage follow reference path.cpp:8-296 and
// volpath.cpp:8-485.
#ifndef MCPT_PATH_CORE_H
#define MCPT_PATH_CORE_H

#include "phase_clock.h"
#include "lights_media.h"
#include "short_stack.h"
#include "traversal.h"
#include "pool_walk.h"

namespace mcpt
{


template <uint32_t kFeatures>
struct Config
{
    static constexpr bool kVolPath = (kFeatures & kFeatVolPath) != 0;
    static constexpr bool kEmitters = (kFeatures & kFeatEmitters) != 0;
    static constexpr bool kAnalytic = (kFeatures & kFeatAnalytic) != 0;
    static constexpr bool kTextures = (kFeatures & kFeatTextures) != 0;
    static constexpr bool kMicrofacet = (kFeatures & kFeatMicrofacet) != 0;
    static constexpr bool kOrdered = (kFeatures & kFeatOrderedWalk) != 0; // which ray query runs (traversal.h)
    static constexpr bool kVote = (kFeatures & kFeatVoteWalk) != 0;       // ... scheduled by wavefront vote
    static constexpr bool kSlivers = (kFeatures & kFeatSlivers) != 0;     // ... with the sliver handling of test_slot
    // which BSDF models beyond diffuse the instantiation compiles (bsdfs.h, bsdf_sample / bsdf_eval)
    static constexpr uint32_t kAllKinds = (1u << kBsdfRoughDiffuse) | (1u << kBsdfConductor) | (1u << kBsdfDielectric) | (1u << kBsdfThinDielectric) | (1u << kBsdfPlastic);
    static constexpr uint32_t kKinds = (kFeatures & kFeatDielectricOnly) ? (1u << kBsdfDielectric)
                                       : (kFeatures & kFeatConductorOnly) ? (1u << kBsdfConductor)
                                       : (kFeatures & kFeatNoTransmission) ? kAllKinds & ~((1u << kBsdfDielectric) | (1u << kBsdfThinDielectric))
                                                                           : kAllKinds;
    static constexpr bool kWide = (kFeatures & kFeatWideWalk) != 0;       // ... on the 4-wide quantised hierarchy
    static constexpr bool kPool = (kFeatures & kFeatPoolWalk) != 0;       // ... as the wavefront-cooperative pool walk (pool_walk.h; device only)
    static constexpr bool kPoolBig = (kFeatures & kFeatPoolBig) != 0;     // ... with 32-bit items, the hierarchy outside LDS
    // ... with MERGED queries: a vertex's last shadow query travels with the next segment's closest query (path_step_merged; the
    // path integrator's pool-walk kernels).  A property of the INSTANTIATION (kFeatPoolMerge), not of a translation unit: which
    // kernels are merged is written where they are instantiated (hip/render_kernel_impl.h: kPB / kPBU, kPM)
    static constexpr bool kPoolDual = kPool && !kVolPath && (kFeatures & kFeatPoolMerge) != 0;
    // distance between a lane's consecutive stack entries (walk_ordered; traversal.h): the workgroup's size on the device
    static constexpr uint32_t kStackStride = (kFeatures & kFeatGroup128) != 0 && kWalkStackStride != 1u ? 128u : kWalkStackStride;
};

// What a vertex leaves behind for the NEXT step's merged query (path_step_merged): its last light's shadow ray and what the sample's
// radiance gains with either answer — and, when the path ended at that vertex's scattering, the sample's radiance so far, to be
// clamped and added to the pixel once the answer is known.
struct PendingShadow
{
    bool shadow;  // a shadow ray waits
    bool finish;  // ... and the sample it belongs to is over: `old_L` is its radiance (st.L belongs to the next sample already)
    V3 origin, dir;
    float t_max;
    V3 add_visible, add_occluded; // throughput x (the vertex's direct light with / without the pending light), path.cpp:98
    V3 old_L;
};

struct LaneCounters
{
    uint32_t closest_rays, shadow_rays, node_tests, prim_tests, shaded_hits, samples;
    uint32_t wave_node_steps, wave_prim_steps;
    // work of the walks of the LAST path_step (model studies on the host build)
    uint32_t last_closest_nodes, last_closest_prims, last_shadow_nodes, last_shadow_prims;
    // ... and the rays themselves (origin, direction, t_max); last_shadow_count of them
    float last_closest_ray[7], last_shadow_ray[7];
    uint32_t last_shadow_count;
    uint32_t last_hit_prim;  // primitive of the last closest hit (kNone: miss) and its distance
    float last_hit_t;
    uint32_t last_shadow_hit; // result of the last shadow walk
};

// Per-lane path state that survives from one step to the next.
struct PathState
{
    uint32_t rng;        // LCG state of the pixel
    uint32_t pixel;      // row-major pixel index
    uint32_t sample;     // next sample to start
    uint32_t depth;      // reference loop counter
    bool alive;          // a path is in flight
    bool primary;        // the ray in flight is the camera ray
    bool in_medium;      // current vertex is a medium scattering event
    uint32_t medium;     // medium of the current medium vertex
    float pdf_sample;    // pdf of the direction that produced the ray in flight
    V3 origin, dir;      // ray in flight
    V3 wo;               // direction towards the previous vertex
    V3 wi;               // sampled direction (points into the vertex)
    V3 throughput, L;    // path weight and radiance of the sample
    V3 pixel_sum;        // sum of clamped samples
    uint32_t *stack;     // this lane's traversal stack (ordered walk): entry k at stack[k * kWalkStackStride]
};

template <class C>
MCPT_HD ShadeTables shade_tables(const DeviceScene &sc)
{
    return ShadeTables{sc.textures, sc.texels, sc.lut_brdf, sc.lut_albedo, !C::kTextures};
}
template <class C>
MCPT_HD LightTables light_tables(const DeviceScene &sc)
{
    return LightTables{sc.textures, sc.texels, sc.env_tables, !C::kTextures};
}

// Work item (position in a draw's tile enumeration: RenderJob) of a pixel — where a packed tile buffer keeps it.
MCPT_HD uint32_t item_of_pixel(uint32_t pixel, uint32_t width, uint32_t tiles_x, uint32_t tile_first, uint32_t tile_stride)
{
    const uint32_t x = pixel % width, y = pixel / width;
    const uint32_t tile = (y >> 3) * tiles_x + (x >> 3);
    return ((tile - tile_first) / tile_stride) * 64u + (y & 7u) * 8u + (x & 7u);
}

// The pre-pass's record of the camera ray of (pixel, sample): primitive (kNone: miss), instance.
MCPT_HD const uint32_t *prehit_record(const DeviceScene &sc, uint32_t pixel, uint32_t sample)
{
    const uint32_t item = item_of_pixel(pixel, static_cast<uint32_t>(sc.camera.width), sc.prehit_tiles_x, sc.prehit_tile_first, sc.prehit_tile_stride);
    return sc.prehit + 2 * (static_cast<size_t>(item) * sc.camera.spp + sample);
}

MCPT_HD void start_pixel(PathState &st, uint32_t pixel)
{
    st.pixel = pixel;
    st.rng = tea4(pixel * 3u, 0); // renderer.cpp:65-66
    st.sample = 0;
    st.alive = false;
    st.pixel_sum = V3{0, 0, 0};
}

// renderer.cpp:68-76: stratified-in-x / van der Corput-in-y jitter, pinhole ray.
// `step` / `independent_seed`: the independent-sample RNG mode (mcpt_renderer_set_rng): the lane renders samples
// s, s + step, ... of the pixel, and each sample starts its own random stream from a PCG hash of (seed, pixel,
// sample) instead of continuing the pixel's — so the samples of a pixel can be spread over lanes.  The defaults
// are the reference: one stream per pixel, every sample.
MCPT_HD void start_sample(const DeviceScene &sc, PathState &st, uint32_t step = 1, bool independent = false, uint32_t seed = 0)
{
    const CameraRec &cam = sc.camera;
    const uint32_t i = st.pixel % static_cast<uint32_t>(cam.width), j = st.pixel / static_cast<uint32_t>(cam.width);
    const uint32_t s = st.sample;
    const float u = s * cam.spp_inv, v = radical_inverse2(s + 1), x = 2.0f * (i + u) / cam.width - 1.0f,
                y = 1.0f - 2.0f * (j + v) / cam.height;
    const V3 look = normalize(from(cam.front) + x * from(cam.dx) + y * from(cam.dy));
    st.origin = from(cam.eye), st.dir = look;
    st.wo = -look;
    st.throughput = V3{1, 1, 1}, st.L = V3{0, 0, 0};
    st.alive = true, st.primary = true, st.in_medium = false;
    st.depth = 0;
    st.pdf_sample = 0;
    st.medium = kNone;
    if (independent)
    {
#if MCPT_LOW_DISCREPANCY_ACTIVE
        st.rng = ld_pack(s, pcg_hash(pcg_hash(seed) + st.pixel)); // (sample index, the pixel's scramble, dimension 0: vecmath.h)
#else
        st.rng = pcg_hash(pcg_hash(pcg_hash(seed) + st.pixel) + s);
#endif
    }
    st.sample += step;
}

// renderer.cpp:77-80: clamp each sample to 1 BEFORE averaging (quirk Q3)
MCPT_HD void finish_sample(PathState &st)
{
    st.pixel_sum += V3{fminf(st.L.x, 1.0f), fminf(st.L.y, 1.0f), fminf(st.L.z, 1.0f)};
    st.alive = false;
}

MCPT_HD bool pixel_done(const DeviceScene &sc, const PathState &st) { return !st.alive && st.sample >= sc.camera.spp; }

MCPT_HD V3 pixel_value(const DeviceScene &sc, const PathState &st) { return st.pixel_sum * sc.camera.spp_inv; }

// One ray query: the ordered walk of the SAH hierarchy, or the reference-order
// walk of the reference's own trees (needed when opacity masks draw random
// numbers during the walk; also the validation mode).
template <class C, bool kAny>
MCPT_HD bool trace(const DeviceScene &sc, uint32_t *stack, Ray &r, uint32_t &rng, HitRaw &hit, TraceStats &ts, bool count)
{
    if (C::kOrdered)
    {
#if MCPT_WAVE_DEVICE
        if (C::kPool) // (`stack` = the wavefront's pool area)
            return count ? walk_pool<kAny, C::kAnalytic, true, C::kPoolBig, C::kSlivers>(sc, stack, true, r, hit, ts) : walk_pool<kAny, C::kAnalytic, false, C::kPoolBig, C::kSlivers>(sc, stack, true, r, hit, ts);
#endif
        if (C::kWide)
            return count ? walk_wide_vote<kAny, C::kAnalytic, true, C::kSlivers>(sc, stack, r, hit, ts)
                         : walk_wide_vote<kAny, C::kAnalytic, false, C::kSlivers>(sc, stack, r, hit, ts);
        if (C::kVote)
            return count ? walk_ordered_vote<kAny, C::kAnalytic, true, C::kSlivers>(sc, stack, r, hit, ts)
                         : walk_ordered_vote<kAny, C::kAnalytic, false, C::kSlivers>(sc, stack, r, hit, ts);
        return count ? walk_ordered<kAny, C::kAnalytic, true, C::kSlivers, C::kStackStride>(sc, stack, r, hit, ts)
                     : walk_ordered<kAny, C::kAnalytic, false, C::kSlivers, C::kStackStride>(sc, stack, r, hit, ts);
    }
    return count ? walk_scene<kAny, C::kAnalytic, C::kTextures, true>(sc, r, rng, hit, ts)
                 : walk_scene<kAny, C::kAnalytic, C::kTextures, false>(sc, r, rng, hit, ts);
}

template <class C>
MCPT_HD bool shadow_walk(const DeviceScene &sc, uint32_t *stack, V3 origin, V3 dir, float t_max, uint32_t &rng,
                         LaneCounters *cnt)
{
    Ray r = make_ray(origin, dir);
    r.t_max = t_max;
    if (cnt)
    {
        const float rec[7] = {origin.x, origin.y, origin.z, dir.x, dir.y, dir.z, t_max};
        for (int k = 0; k < 7; ++k)
            cnt->last_shadow_ray[k] = rec[k];
        ++cnt->last_shadow_count;
    }
    HitRaw dummy;
    TraceStats ts{0, 0, 0, 0};
    const bool hit = trace<C, true>(sc, stack, r, rng, dummy, ts, cnt != nullptr);
    if (cnt)
    {
        ++cnt->shadow_rays, cnt->node_tests += ts.node_tests, cnt->prim_tests += ts.prim_tests;
        cnt->wave_node_steps += ts.wave_node_steps, cnt->wave_prim_steps += ts.wave_prim_steps;
        cnt->last_shadow_nodes += ts.node_tests, cnt->last_shadow_prims += ts.prim_tests;
        cnt->last_shadow_hit = hit ? 1u : 0u;
    }
    return hit;
}

MCPT_HD BsdfQuery query_at(const Surface &s, V3 wo, V3 facing)
{
    BsdfQuery q;
    q.valid = false, q.pdf = 0;
    q.uv = s.uv, q.wo = wo, q.wi = V3{0, 0, 0};
    q.inside = s.inside, q.normal = s.normal, q.tangent = s.tangent, q.bitangent = s.bitangent;
    q.attenuation = V3{0, 0, 0};
    if (dot(facing, s.normal) < 0.0f) // path.cpp:253-257, 282-286
    {
        q.inside = !q.inside;
        q.normal = -q.normal;
    }
    return q;
}

// path.cpp:238-266.  has_bsdf == false: pass-through surface, weight 1.
template <class C, uint32_t kOnly = 0>
MCPT_HD BsdfQuery eval_at(const DeviceScene &sc, const Surface &s, uint32_t bsdf, V3 wi, V3 wo)
{
    BsdfQuery q = query_at(s, wo, -wi);
    q.wi = wi;
    if (bsdf != kNone)
        bsdf_eval<C::kMicrofacet, kOnly, C::kKinds>(shade_tables<C>(sc), sc.bsdfs[bsdf], q);
    else
        q.pdf = 1, q.attenuation = V3{1, 1, 1}, q.valid = true;
    return q;
}

MCPT_HD uint32_t medium_on_side(const DeviceScene &sc, const Surface &s, bool hit_valid, V3 w)
{
    // volpath.cpp:44-45: an invalid hit has a zero normal and invalid ids
    if (!hit_valid)
        return kNone;
    const InstanceRec &rec = sc.instances[s.inst];
    const bool inside = dot(w, s.normal) > 0 ? s.inside : !s.inside;
    return inside ? rec.medium_int : rec.medium_ext;
}

MCPT_HD float area_light_pdf(const DeviceScene &sc, uint32_t light, uint32_t inst, float distance, float cos_light)
{
    const float pdf_area = (sc.light_cdf[light + 1] - sc.light_cdf[light]) * sc.instances[inst].pdf_area;
    return pdf_area * sqr(distance) / cos_light;
}

// Next-event estimation at a surface vertex (path.cpp:138-236,
// volpath.cpp:247-375) or at a medium vertex (volpath.cpp:377-485).
template <class C>
MCPT_HD V3 connect_lights(const DeviceScene &sc, uint32_t *stack, bool at_medium, const Surface &s, V3 position, uint32_t medium_id,
                          V3 wo, uint32_t &rng, LaneCounters *cnt)
{
    V3 L = V3{0, 0, 0};
    const LightTables LT = light_tables<C>(sc);
    const uint32_t bsdf = at_medium ? kNone : sc.instances[s.inst].bsdf;
    // medium the connection travels through
    uint32_t conn_medium = kNone;
    if (C::kVolPath)
        conn_medium = at_medium ? medium_id : (sc.integrator.volpath ? medium_on_side(sc, s, true, wo) : kNone);

    const bool vol = C::kVolPath && sc.integrator.volpath != 0;

    // Transmittance `tr`, scattering value `att` and scattering pdf of a
    // connection arriving along wi over `distance`; false = no contribution.
    // Check order as in the reference: facing test, medium, BSDF (surface
    // vertex) / medium, phase function (medium vertex).
    auto weigh = [&](V3 wi, float distance, V3 &tr, V3 &att, float &pdf) -> bool
    {
        tr = V3{1.0f, 1.0f, 1.0f};
        if (at_medium)
        {
            MediumEvent m = medium_event_init();
            m.distance = distance;
            medium_transmittance(sc.media[medium_id], m);
            if (!m.valid)
                return false;
            tr = m.attenuation / m.pdf;
            PhaseQuery p;
            p.wi = wi, p.wo = wo;
            phase_eval(sc.media[medium_id], p);
            if (!p.valid)
                return false;
            att = p.attenuation, pdf = p.pdf;
            return true;
        }
        if (dot(-wi, s.normal) < kEpsFloat)
            return false;
        if (C::kVolPath && conn_medium != kNone)
        {
            MediumEvent m = medium_event_init();
            m.distance = distance;
            medium_transmittance(sc.media[conn_medium], m);
            if (!m.valid)
                return false;
            tr = m.attenuation / m.pdf;
        }
        const BsdfQuery q = eval_at<C>(sc, s, bsdf, wi, wo);
        if (!q.valid)
            return false;
        att = q.attenuation, pdf = q.pdf;
        return true;
    };

    if (C::kEmitters)
    {
        for (uint32_t k = 0; k < sc.integrator.n_emitters; ++k)
        {
            const EmitterRec &e = sc.emitters[k];
            const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
            const LightSample ls = emitter_sample(LT, e, position, xi0, xi1);
            phase_mark(kPhaseEmitter);
            const bool occluded = shadow_walk<C>(sc, stack, position, -ls.wi, ls.distance - kEpsDistance, rng, cnt);
            phase_mark(kPhaseShadow);
            if (occluded)
                continue;
            V3 tr, att;
            float pdf;
            if (!weigh(ls.wi, ls.distance, tr, att, pdf))
                continue;
            float pdf_direct = 0.0f;
            const V3 radiance = ls.harsh ? emitter_eval_sample(LT, e, ls) : emitter_eval_sample_pdf(LT, e, ls, pdf_direct);
            if (ls.harsh)
            {
                // path.cpp:170 / volpath.cpp:297,413 (tr == 1 exactly without a medium)
                L += vol ? radiance * tr * att : radiance * att;
            }
            else
            {
                if (pdf_direct > kEpsFloat)
                {
                    const float w = power_heuristic(pdf_direct, pdf);
                    if (vol) // volpath.cpp:305-306, 421-422
                        L += w * radiance * tr * att / pdf_direct;
                    else // path.cpp:178
                        L += w * radiance * (att / pdf_direct);
                }
            }
        }
    }

    phase_mark(kPhaseWeigh);
    if (sc.integrator.n_area_lights != 0)
    {
        const float xi_pick = lcg_next(rng);
        const uint32_t light = cdf_search(sc.integrator.n_area_lights + 1, sc.light_cdf, xi_pick) - 1;
        const uint32_t inst = sc.light_inst[light];
        const float xi0 = lcg_next(rng), xi1 = lcg_next(rng), xi2 = lcg_next(rng);
        const LightPoint lp = sample_instance<C::kAnalytic>(sc, inst, xi0, xi1, xi2);
        const V3 d = position - lp.position;
        const float distance = length(d);
        // the shadow ray starts ON THE LIGHT and travels to the shading point
        phase_mark(kPhaseAreaLight);
        const bool occluded = shadow_walk<C>(sc, stack, lp.position, normalize(d), distance - kEpsDistance, rng, cnt);
        phase_mark(kPhaseShadow);
        if (occluded)
            return L;
        const V3 wi = normalize(d);
        const float cos_light = dot(wi, lp.normal);
        if (cos_light < kEpsFloat)
            return L;
        V3 tr, att;
        float pdf;
        if (!weigh(wi, distance, tr, att, pdf))
            return L;
        const float pdf_direct = area_light_pdf(sc, light, inst, distance, cos_light),
                    w = power_heuristic(pdf_direct, pdf);
        const V3 radiance = texture_color(sc.textures, sc.texels, sc.bsdfs[sc.instances[inst].bsdf].tex0, lp.uv, !C::kTextures);
        if (vol) // volpath.cpp:371-372, 481-482
            L += w * (radiance * tr * att / pdf_direct);
        else // path.cpp:232
            L += w * radiance * (att / pdf_direct);
    }
    return L;
}

// One step of a lane whose path is alive: extend the ray in flight, resolve
// what it found, run the roulette, connect to the lights, scatter.  On return
// either st.alive is still true (st.origin/st.dir hold the next ray) or the
// sample has been finished (st.alive == false).
// The step in two halves, so that a kernel can regroup paths between the ray query and the shading (the class-sorted
// kernel, hip/sorted_kernel.hip).  path_extend: the closest-hit query of the ray in flight (ray, raw, return value =
// hit_valid; ray.t_max = the distance).  path_shade: everything after it.  path_step = one after the other.
template <class C>
MCPT_HD bool path_extend(const DeviceScene &sc, PathState &st, LaneCounters *cnt, Ray &ray, HitRaw &raw)
{
    ray = make_ray(st.origin, st.dir);
    TraceStats ts{0, 0, 0, 0};
    bool hit_valid;
    const bool known = C::kOrdered && st.primary && sc.prehit != nullptr; // the pre-pass traced this camera ray
    if (known)
    {
        const uint32_t *rec = prehit_record(sc, st.pixel, st.sample - sc.prehit_step); // (start_sample advanced it)
        const uint32_t prim = rec[0];
        hit_valid = prim != kNone;
        if (hit_valid)
            hit_from_record<C::kAnalytic>(sc, rec[1], prim, ray, raw);
    }
    else
        hit_valid = trace<C, false>(sc, st.stack, ray, st.rng, raw, ts, cnt != nullptr);
    if (cnt && !known)
    {
        ++cnt->closest_rays, cnt->node_tests += ts.node_tests, cnt->prim_tests += ts.prim_tests;
        cnt->wave_node_steps += ts.wave_node_steps, cnt->wave_prim_steps += ts.wave_prim_steps;
        cnt->last_closest_nodes = ts.node_tests, cnt->last_closest_prims = ts.prim_tests;
        cnt->last_shadow_nodes = cnt->last_shadow_prims = 0;
        const float rec[7] = {st.origin.x, st.origin.y, st.origin.z, st.dir.x, st.dir.y, st.dir.z, kMaxFloat};
        for (int k = 0; k < 7; ++k)
            cnt->last_closest_ray[k] = rec[k];
        cnt->last_shadow_count = 0;
        cnt->last_hit_prim = hit_valid ? raw.prim : kNone, cnt->last_hit_t = ray.t_max;
    }
    return hit_valid;
}

// path_resolve: what the ray found (surface frame, primary miss, free-flight sampling of the medium it crossed, escape,
// back face, light) and the roulette; afterwards either the sample is finished (st.alive == false) or the path stands at
// its next vertex — a medium scattering event (st.in_medium, st.origin) or the surface point `surf` — and
// path_connect_scatter does the rest of the step there.  path_shade = one after the other.
// (kPointOnly: `surf` gets the half of the record this function reads — instance, side, position, shading normal — and the caller
//  completes it with make_surface_part<..., 2> for the lanes that go on from a surface vertex: instantiations without textures only)
template <class C, bool kPointOnly = false>
MCPT_HD void path_resolve(const DeviceScene &sc, PathState &st, LaneCounters *cnt, const Ray &ray, const HitRaw &raw, bool hit_valid,
                          Surface &surf)
{
    const IntegratorRec &ig = sc.integrator;
    const bool vol = C::kVolPath && ig.volpath != 0;
    const LightTables LT = light_tables<C>(sc);
    if (hit_valid)
    {
        if (kPointOnly)
        {
            surf.uv = V2{0, 0}, surf.tangent = surf.bitangent = V3{0, 0, 0};
            make_surface_part<C::kAnalytic, C::kTextures, kPointOnly ? 1 : 0>(sc, raw, surf);
        }
        else
            surf = make_surface<C::kAnalytic, C::kTextures>(sc, ray, raw);
        if (cnt)
            ++cnt->shaded_hits;
    }
    else
    {
        surf.inside = false, surf.inst = 0, surf.uv = V2{0, 0};
        surf.position = surf.normal = surf.tangent = surf.bitangent = V3{0, 0, 0};
    }
    phase_mark(kPhaseSurface);

    // ---- resolve -------------------------------------------------------------
    if (st.primary && !hit_valid)
    {
        // path.cpp:22-35 / volpath.cpp:22-35: environment AND sun on a primary miss
        if (ig.id_envmap != kNone)
            st.L += emitter_eval_dir(LT, sc.emitters[ig.id_envmap], st.dir);
        if (ig.id_sun != kNone)
            st.L += emitter_eval_dir(LT, sc.emitters[ig.id_sun], st.dir);
        finish_sample(st);
        return;
    }

    if (vol)
    {
        // free-flight sampling along the segment just traced (volpath.cpp:41-63,
        // 115-139, 160-186)
        const bool from_medium = st.in_medium;
        const uint32_t id = from_medium ? st.medium : medium_on_side(sc, surf, hit_valid, st.primary ? st.wo : st.wi);
        if (from_medium)
            st.in_medium = false;
        if (id != kNone)
        {
            MediumEvent m = medium_event_init();
            medium_sample_distance(sc.media[id], ray.t_max, st.rng, m);
            if (m.valid)
            {
                st.throughput *= m.attenuation / m.pdf;
                if (m.scattered)
                {
                    st.in_medium = true;
                    st.medium = id;
                    st.origin = ray.origin + ray.dir * m.distance; // the medium vertex
                }
            }
        }
    }

    phase_mark(kPhaseMedium);
    const uint32_t bsdf = hit_valid ? sc.instances[surf.inst].bsdf : kNone;
    if (!st.in_medium)
    {
        if (!hit_valid)
        {
            // secondary ray escaped: MIS-weighted environment only (path.cpp:79-94)
            if (ig.id_envmap != kNone)
            {
                const EmitterRec &env = sc.emitters[ig.id_envmap];
                float pdf_direct;
                const V3 radiance = emitter_eval_dir_pdf(LT, env, -st.wi, pdf_direct);
                const float w = power_heuristic(st.pdf_sample, pdf_direct);
                st.L += w * st.throughput * radiance;
            }
            finish_sample(st);
            return;
        }
        if (bsdf != kNone)
        {
            const BsdfRec &b = sc.bsdfs[bsdf];
            if (surf.inside && !b.twosided)
            {
                // back of a one-sided surface absorbs (path.cpp:43-46, 101-104);
                // for a primary ray the sample is black (L is still 0)
                finish_sample(st);
                return;
            }
            if (b.kind == kBsdfAreaLight)
            {
                const V3 radiance = texture_color(sc.textures, sc.texels, b.tex0, surf.uv, !C::kTextures);
                if (st.primary)
                {
                    if (!ig.hide_emitters) // path.cpp:47-53
                        st.L = radiance;
                }
                else
                {
                    const float cos_light = dot(st.wi, surf.normal); // path.cpp:105-124
                    if (cos_light >= kEpsFloat)
                    {
                        const float pdf_direct = area_light_pdf(sc, sc.instances[surf.inst].area_light, surf.inst,
                                                                ray.t_max, cos_light),
                                    w = power_heuristic(st.pdf_sample, pdf_direct);
                        st.L += w * st.throughput * radiance;
                    }
                }
                finish_sample(st);
                return;
            }
        }
        if (!st.primary)
        {
            st.wo = st.wi; // path.cpp:127-132
            if (st.depth >= ig.depth_rr)
                st.throughput *= ig.rr_scale; // multiplies by pdf_rr: reference quirk Q2
        }
    }

    // ---- roulette: the reference's for-loop header (path.cpp:57-59) -----------
    st.depth = st.primary ? 1u : st.depth + 1u;
    st.primary = false;
    if (!(st.depth < ig.depth_rr || (st.depth < ig.depth_max && lcg_next(st.rng) < ig.pdf_rr)))
    {
        finish_sample(st);
        return;
    }
}

template <class C>
MCPT_HD void path_connect_scatter(const DeviceScene &sc, PathState &st, LaneCounters *cnt, const Surface &surf)
{
    const bool vol = C::kVolPath && sc.integrator.volpath != 0;
    const uint32_t bsdf = st.in_medium ? kNone : sc.instances[surf.inst].bsdf; // (a surface vertex is a valid hit)
    // ---- connect -------------------------------------------------------------
    const V3 vertex = st.in_medium ? st.origin : surf.position;
    st.L += st.throughput * connect_lights<C>(sc, st.stack, st.in_medium, surf, vertex, st.medium, st.wo, st.rng, cnt);
    phase_mark(kPhaseWeigh);

    // ---- scatter -------------------------------------------------------------
    if (vol && st.in_medium)
    {
        PhaseQuery p; // volpath.cpp:96-110
        p.wo = st.wo;
        phase_sample(sc.media[st.medium], st.rng, p);
        phase_mark(kPhasePhase);
        if (!p.valid)
        {
            finish_sample(st);
            return;
        }
        st.wi = p.wi;
        st.throughput *= p.attenuation / p.pdf;
        st.pdf_sample = p.pdf;
    }
    else
    {
        BsdfQuery q = query_at(surf, st.wo, st.wo); // path.cpp:268-296
        if (bsdf != kNone)
        {
            bsdf_sample<C::kMicrofacet, 0, C::kKinds>(shade_tables<C>(sc), sc.bsdfs[bsdf], st.rng, q);
        }
        else
        {
            // a shape without BSDF is a pass-through surface (quirk Q8)
            q.wi = st.wo, q.pdf = 1.0f, q.attenuation = V3{1.0f, 1.0f, 1.0f}, q.valid = true;
        }
        phase_mark(kPhaseBsdf);
        if (!q.valid)
        {
            finish_sample(st);
            return;
        }
        st.wi = q.wi;
        st.pdf_sample = q.pdf;
        st.throughput *= q.attenuation / q.pdf;
        st.origin = surf.position;
    }
    if (max_component(st.throughput) < kEps)
    {
        finish_sample(st);
        return;
    }
    st.dir = -st.wi;
}

template <class C>
MCPT_HD void path_shade(const DeviceScene &sc, PathState &st, LaneCounters *cnt, const Ray &ray, const HitRaw &raw, bool hit_valid)
{
    Surface surf;
    path_resolve<C>(sc, st, cnt, ray, raw, hit_valid, surf);
    if (st.alive)
        path_connect_scatter<C>(sc, st, cnt, surf);
}

template <class C>
MCPT_HD void path_step(const DeviceScene &sc, PathState &st, LaneCounters *cnt)
{
    // ---- extend --------------------------------------------------------------
    Ray ray;
    HitRaw raw;
    const bool hit_valid = path_extend<C>(sc, st, cnt, ray, raw);
    path_shade<C>(sc, st, cnt, ray, raw, hit_valid);
}

#if MCPT_WAVE_CODE
// ---- the step with UNIFORM ray queries (pool walk, pool_walk.h) -----------------------------------------------
// path_step for the kernels whose ray queries are the wavefront-cooperative pool walk: every lane of the wavefront makes
// every query call of the step, with a flag saying whether it brings a ray — a lane without a path (`has_path` false), or
// whose sample ended before the query, works on the other lanes' rays.  Statement for statement path_extend /
// path_resolve / connect_lights / the scatter of path_connect_scatter: the random numbers of a vertex are drawn in the
// reference's order (every draw of a light precedes its shadow query, nothing after a query draws: path.cpp:144-205).
template <class C, bool kAny>
__device__ __forceinline__ bool trace_uniform(const DeviceScene &sc, uint32_t *pool, bool has_ray, Ray &r, HitRaw &hit, TraceStats &ts, bool count)
{
    return count ? walk_pool<kAny, C::kAnalytic, true, C::kPoolBig, C::kSlivers>(sc, pool, has_ray, r, hit, ts) : walk_pool<kAny, C::kAnalytic, false, C::kPoolBig, C::kSlivers>(sc, pool, has_ray, r, hit, ts);
}

template <class C>
__device__ __forceinline__ bool shadow_walk_uniform(const DeviceScene &sc, uint32_t *pool, bool has_ray, V3 origin, V3 dir, float t_max, LaneCounters *cnt)
{
    Ray r = make_ray(origin, dir);
    r.t_max = t_max;
    HitRaw dummy;
    TraceStats ts{0, 0, 0, 0};
    const bool hit = trace_uniform<C, true>(sc, pool, has_ray, r, dummy, ts, cnt != nullptr);
    if (cnt)
    {
        cnt->shadow_rays += has_ray ? 1u : 0u, cnt->node_tests += ts.node_tests, cnt->prim_tests += ts.prim_tests;
        cnt->wave_node_steps += ts.wave_node_steps, cnt->wave_prim_steps += ts.wave_prim_steps;
    }
    return hit;
}

// connect_lights with the shadow queries made by every lane; `active`: this lane stands at a vertex (the other arguments
// mean something only then).
template <class C>
__device__ __forceinline__ V3 connect_lights_uniform(const DeviceScene &sc, uint32_t *pool, bool active, bool at_medium, const Surface &s, V3 position,
                                                     uint32_t medium_id, V3 wo, uint32_t &rng, LaneCounters *cnt)
{
    V3 L = V3{0, 0, 0};
    const LightTables LT = light_tables<C>(sc);
    const uint32_t bsdf = (!active || at_medium) ? kNone : sc.instances[s.inst].bsdf;
    uint32_t conn_medium = kNone; // medium the connection travels through
    if (C::kVolPath && active)
        conn_medium = at_medium ? medium_id : (sc.integrator.volpath ? medium_on_side(sc, s, true, wo) : kNone);
    const bool vol = C::kVolPath && sc.integrator.volpath != 0;
    auto weigh = [&](V3 wi, float distance, V3 &tr, V3 &att, float &pdf) -> bool
    {
        tr = V3{1.0f, 1.0f, 1.0f};
        if (C::kVolPath && at_medium)
        {
            MediumEvent m = medium_event_init();
            m.distance = distance;
            medium_transmittance(sc.media[medium_id], m);
            if (!m.valid)
                return false;
            tr = m.attenuation / m.pdf;
            PhaseQuery p;
            p.wi = wi, p.wo = wo;
            phase_eval(sc.media[medium_id], p);
            if (!p.valid)
                return false;
            att = p.attenuation, pdf = p.pdf;
            return true;
        }
        if (dot(-wi, s.normal) < kEpsFloat)
            return false;
        if (C::kVolPath && conn_medium != kNone)
        {
            MediumEvent m = medium_event_init();
            m.distance = distance;
            medium_transmittance(sc.media[conn_medium], m);
            if (!m.valid)
                return false;
            tr = m.attenuation / m.pdf;
        }
        const BsdfQuery q = eval_at<C>(sc, s, bsdf, wi, wo);
        if (!q.valid)
            return false;
        att = q.attenuation, pdf = q.pdf;
        return true;
    };
    if (C::kEmitters)
    {
        for (uint32_t k = 0; k < sc.integrator.n_emitters; ++k)
        {
            const EmitterRec &e = sc.emitters[k];
            LightSample ls{};
            if (active)
            {
                const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
                ls = emitter_sample(LT, e, position, xi0, xi1);
            }
            phase_mark(kPhaseEmitter, active);
            const bool occluded = shadow_walk_uniform<C>(sc, pool, active, position, -ls.wi, ls.distance - kEpsDistance, cnt);
            phase_mark(kPhaseShadow, active);
            V3 tr, att;
            float pdf;
            if (!active || occluded || !weigh(ls.wi, ls.distance, tr, att, pdf))
                continue;
            float pdf_direct = 0.0f;
            const V3 radiance = ls.harsh ? emitter_eval_sample(LT, e, ls) : emitter_eval_sample_pdf(LT, e, ls, pdf_direct);
            if (ls.harsh)
                L += vol ? radiance * tr * att : radiance * att; // path.cpp:170 / volpath.cpp:297,413
            else
            {
                if (pdf_direct > kEpsFloat)
                {
                    const float w = power_heuristic(pdf_direct, pdf);
                    if (vol) // volpath.cpp:305-306, 421-422
                        L += w * radiance * tr * att / pdf_direct;
                    else // path.cpp:178
                        L += w * radiance * (att / pdf_direct);
                }
            }
        }
    }
    if (sc.integrator.n_area_lights != 0)
    {
        uint32_t light = 0, inst = 0;
        LightPoint lp{};
        V3 d = V3{0, 0, 1};
        float distance = 0.0f;
        if (active)
        {
            const float xi_pick = lcg_next(rng);
            light = cdf_search(sc.integrator.n_area_lights + 1, sc.light_cdf, xi_pick) - 1;
            inst = sc.light_inst[light];
            const float xi0 = lcg_next(rng), xi1 = lcg_next(rng), xi2 = lcg_next(rng);
            lp = sample_instance<C::kAnalytic>(sc, inst, xi0, xi1, xi2);
            d = position - lp.position;
            distance = length(d);
        }
        // the shadow ray starts ON THE LIGHT and travels to the shading point
        phase_mark(kPhaseAreaLight, active);
        const bool occluded = shadow_walk_uniform<C>(sc, pool, active, lp.position, normalize(d), distance - kEpsDistance, cnt);
        phase_mark(kPhaseShadow, active);
        if (active && !occluded)
        {
            const V3 wi = normalize(d);
            const float cos_light = dot(wi, lp.normal);
            V3 tr, att;
            float pdf;
            if (!(cos_light < kEpsFloat) && weigh(wi, distance, tr, att, pdf))
            {
                const float pdf_direct = area_light_pdf(sc, light, inst, distance, cos_light), w = power_heuristic(pdf_direct, pdf);
                const V3 radiance = texture_color(sc.textures, sc.texels, sc.bsdfs[sc.instances[inst].bsdf].tex0, lp.uv, !C::kTextures);
                if (vol) // volpath.cpp:371-372, 481-482
                    L += w * (radiance * tr * att / pdf_direct);
                else // path.cpp:232
                    L += w * radiance * (att / pdf_direct);
            }
        }
    }
    return L;
}

// path_extend / path_connect_scatter with every lane making the query calls (the class-sorted kernel's two halves).
template <class C>
__device__ __forceinline__ bool path_extend_uniform(const DeviceScene &sc, PathState &st, LaneCounters *cnt, bool has_path, Ray &ray, HitRaw &raw)
{
    ray = make_ray(has_path ? st.origin : V3{0, 0, 0}, has_path ? st.dir : V3{0, 0, 1});
    raw.inst = raw.prim = 0, raw.a = raw.b = raw.c = 0.0f, raw.inside = false;
    TraceStats ts{0, 0, 0, 0};
    const bool known = has_path && st.primary && sc.prehit != nullptr; // the pre-pass traced this camera ray
    bool hit_valid = false;
    if (known)
    {
        const uint32_t *rec = prehit_record(sc, st.pixel, st.sample - sc.prehit_step);
        hit_valid = rec[0] != kNone;
        if (hit_valid)
            hit_from_record<C::kAnalytic>(sc, rec[1], rec[0], ray, raw);
    }
    const bool traced = trace_uniform<C, false>(sc, st.stack, has_path && !known, ray, raw, ts, cnt != nullptr);
    if (cnt)
    {
        cnt->closest_rays += has_path && !known ? 1u : 0u, cnt->node_tests += ts.node_tests, cnt->prim_tests += ts.prim_tests;
        cnt->wave_node_steps += ts.wave_node_steps, cnt->wave_prim_steps += ts.wave_prim_steps;
    }
    return known ? hit_valid : traced;
}

template <class C>
__device__ __forceinline__ void path_connect_scatter_uniform(const DeviceScene &sc, PathState &st, LaneCounters *cnt, const Surface &surf, bool active)
{
    const bool vol = C::kVolPath && sc.integrator.volpath != 0;
    const bool at_medium = active && st.in_medium;
    // ---- connect ----
    const V3 vertex = at_medium ? st.origin : surf.position;
    const V3 direct = connect_lights_uniform<C>(sc, st.stack, active, at_medium, surf, vertex, st.medium, st.wo, st.rng, cnt);
    phase_mark(kPhaseWeigh, active);
    if (!active)
        return;
    st.L += st.throughput * direct;
    // ---- scatter (path_connect_scatter) ----
    const uint32_t bsdf = st.in_medium ? kNone : sc.instances[surf.inst].bsdf;
    if (vol && st.in_medium)
    {
        PhaseQuery p; // volpath.cpp:96-110
        p.wo = st.wo;
        phase_sample(sc.media[st.medium], st.rng, p);
        if (!p.valid)
        {
            finish_sample(st);
            return;
        }
        st.wi = p.wi;
        st.throughput *= p.attenuation / p.pdf;
        st.pdf_sample = p.pdf;
    }
    else
    {
        BsdfQuery q = query_at(surf, st.wo, st.wo); // path.cpp:268-296
        if (bsdf != kNone)
            bsdf_sample<C::kMicrofacet, 0, C::kKinds>(shade_tables<C>(sc), sc.bsdfs[bsdf], st.rng, q);
        else
            q.wi = st.wo, q.pdf = 1.0f, q.attenuation = V3{1.0f, 1.0f, 1.0f}, q.valid = true; // pass-through surface (quirk Q8)
        phase_mark(kPhaseBsdf);
        if (!q.valid)
        {
            finish_sample(st);
            return;
        }
        st.wi = q.wi;
        st.pdf_sample = q.pdf;
        st.throughput *= q.attenuation / q.pdf;
        st.origin = surf.position;
    }
    if (max_component(st.throughput) < kEps)
    {
        finish_sample(st);
        return;
    }
    st.dir = -st.wi;
}

// THE NEXT SAMPLE IN THE SAME STEP (round 6; scenes with the camera-ray pre-pass).  A sample that ends in path_resolve — its ray left
// the scene, hit a light, was stopped by the roulette — used to leave its lane idle for the rest of the step (connect, shadow query,
// scatter: half of a step's instructions) and take the whole next step for its successor's FIRST vertex, whose ray query the
// pre-pass has already answered.  Here the lane starts its pixel's next sample on the spot — the same random stream, the samples still
// one after the other — reads the camera ray's hit from the pre-pass, resolves it and joins this step's connect / scatter with its
// first vertex; a camera ray that misses ends that sample too, and the lane tries the next (kRegenerateRounds per step at most: the
// other lanes wait meanwhile).  matpreview: the lanes that reach connect / scatter were 36-42 of 64.  Nothing a path computes changes.
#ifndef MCPT_REGENERATE_ROUNDS
#define MCPT_REGENERATE_ROUNDS 4
#endif
constexpr uint32_t kRegenerateRounds = MCPT_REGENERATE_ROUNDS;
struct SampleStart // start_sample's arguments (RenderJob: sample_split, independent_samples, rng_seed)
{
    uint32_t step;
    bool independent;
    uint32_t seed;
};
// path_resolve for every lane of the step, with the regeneration above around it:
//   1. lanes whose ray found NOTHING are resolved first — no surface frame to build: the escape's radiance, the sample ends (or stands at a
//      medium vertex) — and start their next sample, whose camera ray the pre-pass has answered; again while that one misses too;
//   2. ONE path_resolve for all lanes that stand at a hit — the surface frames (a triangle's: 36 floats from memory) built once, with
//      the regenerated lanes among them (phase clock, matpreview: that phase ran twice a step at 30 of 64 lanes and was 15 % of the time);
//   3. lanes that ended THERE (a light, a back face, the roulette) start their next sample and resolve it, as before.
template <class C>
__device__ __forceinline__ void resolve_and_regenerate(const DeviceScene &sc, PathState &st, LaneCounters *cnt, bool has_path, const SampleStart &how, Ray &ray, HitRaw &raw,
                                                       bool hit_valid, Surface &surf)
{
    const bool regenerate = kRegenerateRounds != 0 && sc.prehit != nullptr; // (uniform)
    // Step 1 only in the diffuse instantiations (dragon/scene.xml's class: three quarters of its camera rays miss, and a miss costs a
    // constant): dragon 92.7 -> 90.3 ms.  The surface-material instantiations lose with it — matpreview 374.7 -> 379.3, 639.8 -> 652.4 ms:
    // their escapes are environment-map lookups, and those first, then the frames, is two latency chains one after the other where one
    // diverged path_resolve overlaps them (EXPERIMENTS R6-17).
    // (-DMCPT_REGEN_FORM=1 / 2, experiment builds: step 1 in every instantiation, with / without step 3)
#ifndef MCPT_REGEN_FORM
#define MCPT_REGEN_FORM 0
#endif
    constexpr bool kEscapedFirst = MCPT_REGEN_FORM != 0 || (!C::kMicrofacet && !C::kTextures);
    constexpr bool kAfterwards = MCPT_REGEN_FORM != 2;
    bool resolved = !has_path;
    auto next_sample = [&]() __attribute__((always_inline))
    {
        start_sample(sc, st, how.step, how.independent, how.seed);
        if (cnt)
            ++cnt->samples;
        ray = make_ray(st.origin, st.dir);
        const uint32_t *rec = prehit_record(sc, st.pixel, st.sample - sc.prehit_step); // (path_extend_uniform's "known" branch)
        hit_valid = rec[0] != kNone;
        if (hit_valid)
            hit_from_record<C::kAnalytic>(sc, rec[1], rec[0], ray, raw);
        resolved = false;
    };
    if (kEscapedFirst && regenerate)
    {
#pragma unroll 1
        for (uint32_t round = 0; round < kRegenerateRounds; ++round)
        {
            const bool escaped = !resolved && !hit_valid;
            if (__ballot(escaped) == 0)
                break;
            if (escaped)
            {
                path_resolve<C>(sc, st, cnt, ray, raw, false, surf);
                resolved = true;
                if (!st.alive && st.sample < sc.camera.spp)
                    next_sample();
            }
        }
    }
    if (!resolved)
    {
        path_resolve<C>(sc, st, cnt, ray, raw, hit_valid, surf);
        resolved = true;
    }
    if (kAfterwards && regenerate)
    {
#pragma unroll 1
        for (uint32_t round = 0; round < kRegenerateRounds; ++round)
        {
            const bool again = has_path && !st.alive && st.sample < sc.camera.spp;
            if (__ballot(again) == 0)
                break;
            if (again)
            {
                next_sample();
                path_resolve<C>(sc, st, cnt, ray, raw, hit_valid, surf);
            }
        }
    }
}

// One step of every lane of the wavefront; `has_path`: the lane's path is alive (st.alive) — the others only help.
template <class C>
__device__ __forceinline__ void path_step_uniform(const DeviceScene &sc, PathState &st, LaneCounters *cnt, bool has_path, const SampleStart &how = SampleStart{1u, false, 0u})
{
    static_assert(C::kOrdered, "ordered walk");
    Ray ray;
    HitRaw raw;
    const bool hit_valid = path_extend_uniform<C>(sc, st, cnt, has_path, ray, raw);
    phase_mark(kPhaseExtend, has_path);
    Surface surf;
    surf.inside = false, surf.inst = 0, surf.uv = V2{0, 0};
    surf.position = surf.normal = surf.tangent = surf.bitangent = V3{0, 0, 0};
    resolve_and_regenerate<C>(sc, st, cnt, has_path, how, ray, raw, hit_valid, surf);
    phase_mark(kPhaseResolve, has_path && st.alive);
    path_connect_scatter_uniform<C>(sc, st, cnt, surf, has_path && st.alive);
    phase_mark(kPhaseScatter, has_path && st.alive);
}

// ---- MERGED QUERIES (round 5) -----------------------------------------------------------------------------------
// The reference traces, at every vertex, one shadow ray per light and then the next segment (path.cpp:144-205, 268-296), and
// nothing between a vertex's last shadow query and the next closest query draws a random number (the walks draw none; the
// scattering's draws come BEFORE it is known whether the light was visible, and the radiance of the light is only ADDED to the
// sample).  So the last shadow ray of a vertex can travel together with the next segment's ray: one pool walk with up to two
// rays per lane (pool_walk.h, kDual) instead of two walks of one — half the query rounds of a path (each a chain of dependent
// steps), twice the items per step.  What the vertex would add to the sample is kept with either answer (PendingShadow) and
// added, in the reference's order of additions, right after the walk — before the next vertex adds anything.  A path that ends
// at the scattering of a vertex with a pending ray is finished (clamped, added to the pixel) one step later, while its lane's
// next sample is already under way; samples are still finished in order.  Lights before the last one are queried on the spot.
template <class C>
__device__ __forceinline__ void connect_lights_merged(const DeviceScene &sc, uint32_t *pool, bool active, const Surface &s, V3 position, V3 wo, V3 throughput,
                                                      uint32_t &rng, V3 &sample_L, PendingShadow &pd, LaneCounters *cnt)
{
    static_assert(!C::kVolPath, "merged queries: the path integrator");
    V3 L = V3{0, 0, 0};
    const LightTables LT = light_tables<C>(sc);
    const uint32_t bsdf = !active ? kNone : sc.instances[s.inst].bsdf;
    auto weigh = [&](V3 wi, V3 &att, float &pdf) -> bool
    {
        if (dot(-wi, s.normal) < kEpsFloat)
            return false;
        const BsdfQuery q = eval_at<C>(sc, s, bsdf, wi, wo);
        if (!q.valid)
            return false;
        att = q.attenuation, pdf = q.pdf;
        return true;
    };
    bool wait = false; // this lane's last light waits for the next walk
    V3 last = V3{0, 0, 0};
    if (C::kEmitters)
    {
        for (uint32_t k = 0; k < sc.integrator.n_emitters; ++k)
        {
            const EmitterRec &e = sc.emitters[k];
            const bool deferred = sc.integrator.n_area_lights == 0 && k + 1 == sc.integrator.n_emitters; // (uniform)
            LightSample ls{};
            if (active)
            {
                const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
                ls = emitter_sample(LT, e, position, xi0, xi1);
            }
            bool occluded = false;
            phase_mark(kPhaseEmitter, active);
            if (!deferred)
            {
                occluded = shadow_walk_uniform<C>(sc, pool, active, position, -ls.wi, ls.distance - kEpsDistance, cnt);
                phase_mark(kPhaseShadow, active);
            }
            V3 att;
            float pdf;
            if (!active || occluded || !weigh(ls.wi, att, pdf))
                continue;
            float pdf_direct = 0.0f;
            const V3 radiance = ls.harsh ? emitter_eval_sample(LT, e, ls) : emitter_eval_sample_pdf(LT, e, ls, pdf_direct);
            V3 c = V3{0, 0, 0};
            bool some = false;
            if (ls.harsh)
                c = radiance * att, some = true; // path.cpp:170
            else
            {
                if (pdf_direct > kEpsFloat)
                    c = power_heuristic(pdf_direct, pdf) * radiance * (att / pdf_direct), some = true; // path.cpp:178
            }
            if (!some)
                continue;
            if (!deferred)
                L += c;
            else
            {
                wait = true, last = c;
                pd.origin = position, pd.dir = -ls.wi, pd.t_max = ls.distance - kEpsDistance;
            }
        }
    }
    if (sc.integrator.n_area_lights != 0)
    {
        if (active)
        {
            const float xi_pick = lcg_next(rng);
            const uint32_t light = cdf_search(sc.integrator.n_area_lights + 1, sc.light_cdf, xi_pick) - 1;
            const uint32_t inst = sc.light_inst[light];
            const float xi0 = lcg_next(rng), xi1 = lcg_next(rng), xi2 = lcg_next(rng);
            const LightPoint lp = sample_instance<C::kAnalytic>(sc, inst, xi0, xi1, xi2);
            const V3 d = position - lp.position;
            const float distance = length(d);
            const V3 wi = normalize(d);
            const float cos_light = dot(wi, lp.normal);
            V3 att;
            float pdf;
            if (!(cos_light < kEpsFloat) && weigh(wi, att, pdf))
            {
                const float pdf_direct = area_light_pdf(sc, light, inst, distance, cos_light), w = power_heuristic(pdf_direct, pdf);
                const V3 radiance = texture_color(sc.textures, sc.texels, sc.bsdfs[sc.instances[inst].bsdf].tex0, lp.uv, !C::kTextures);
                wait = true, last = w * radiance * (att / pdf_direct); // path.cpp:232
                // the shadow ray starts ON THE LIGHT and travels to the shading point
                pd.origin = lp.position, pd.dir = wi, pd.t_max = distance - kEpsDistance;
            }
        }
    }
    if (!active)
        return;
    // path.cpp:98: L += throughput * direct, direct = (((0 + c_1) + c_2) ... + c_last): the same sums and products, the last term
    // with both answers
    if (wait)
    {
        pd.shadow = true;
        pd.add_occluded = throughput * L, pd.add_visible = throughput * (L + last);
    }
    else
        sample_L += throughput * L;
}

// One step of every lane of the wavefront with merged queries.  `has_path`: the lane's path is alive; a lane may also only have a
// pending shadow ray (its sample's path ended at the last vertex), or nothing (a helper of the others' rays).
template <class C>
__device__ __forceinline__ void path_step_merged(const DeviceScene &sc, PathState &st, PendingShadow &pd, LaneCounters *cnt, bool has_path, const SampleStart &how = SampleStart{1u, false, 0u})
{
    static_assert(C::kOrdered && C::kPoolDual, "merged queries run on the pool walk");
    // ---- the walk: this segment's closest query + the last vertex's pending shadow query ----
    Ray ray = make_ray(has_path ? st.origin : V3{0, 0, 0}, has_path ? st.dir : V3{0, 0, 1});
    HitRaw raw;
    raw.inst = raw.prim = 0, raw.a = raw.b = raw.c = 0.0f, raw.inside = false;
    TraceStats ts{0, 0, 0, 0};
    const bool known = has_path && st.primary && sc.prehit != nullptr; // the pre-pass traced this camera ray
    bool hit_valid = false;
    if (known)
    {
        const uint32_t *rec = prehit_record(sc, st.pixel, st.sample - sc.prehit_step);
        hit_valid = rec[0] != kNone;
        if (hit_valid)
            hit_from_record<C::kAnalytic>(sc, rec[1], rec[0], ray, raw);
    }
    Ray shadow = make_ray(pd.shadow ? pd.origin : V3{0, 0, 0}, pd.shadow ? pd.dir : V3{0, 0, 1});
    shadow.t_max = pd.t_max;
    bool occluded = false;
    const bool traced = cnt ? walk_pool<false, C::kAnalytic, true, C::kPoolBig, C::kSlivers, true>(sc, st.stack, has_path && !known, ray, raw, ts, pd.shadow, &shadow, &occluded)
                            : walk_pool<false, C::kAnalytic, false, C::kPoolBig, C::kSlivers, true>(sc, st.stack, has_path && !known, ray, raw, ts, pd.shadow, &shadow, &occluded);
    if (cnt)
    {
        cnt->closest_rays += has_path && !known ? 1u : 0u, cnt->shadow_rays += pd.shadow ? 1u : 0u;
        cnt->node_tests += ts.node_tests, cnt->prim_tests += ts.prim_tests;
        cnt->wave_node_steps += ts.wave_node_steps, cnt->wave_prim_steps += ts.wave_prim_steps;
    }
    hit_valid = known ? hit_valid : traced;
    phase_mark(kPhaseExtend, has_path);
    // ---- what the last vertex adds to its sample, now that its light's visibility is known ----
    if (pd.shadow)
    {
        const V3 add = occluded ? pd.add_occluded : pd.add_visible;
        if (pd.finish)
        {
            pd.old_L += add;
            st.pixel_sum += V3{fminf(pd.old_L.x, 1.0f), fminf(pd.old_L.y, 1.0f), fminf(pd.old_L.z, 1.0f)}; // finish_sample, one step late
        }
        else
            st.L += add;
        pd.shadow = pd.finish = false;
    }
    // ---- resolve this segment's vertex, connect (the last light waits), scatter ----
    Surface surf;
    surf.inside = false, surf.inst = 0, surf.uv = V2{0, 0};
    surf.position = surf.normal = surf.tangent = surf.bitangent = V3{0, 0, 0};
    resolve_and_regenerate<C>(sc, st, cnt, has_path, how, ray, raw, hit_valid, surf); // (the pending ray above has been answered: nothing of the ended sample is open)
    const bool active = has_path && st.alive;
    phase_mark(kPhaseResolve, active);
    connect_lights_merged<C>(sc, st.stack, active, surf, surf.position, st.wo, st.throughput, st.rng, st.L, pd, cnt);
    phase_mark(kPhaseWeigh, active);
    if (!active)
        return;
    // (scatter of path_connect_scatter, path.cpp:268-296; a sample that ends here with a light pending is finished after the next walk)
    auto end_sample = [&]()
    {
        if (pd.shadow)
            pd.finish = true, pd.old_L = st.L, st.alive = false;
        else
            finish_sample(st);
    };
    const uint32_t bsdf = sc.instances[surf.inst].bsdf;
    BsdfQuery q = query_at(surf, st.wo, st.wo);
    if (bsdf != kNone)
        bsdf_sample<C::kMicrofacet, 0, C::kKinds>(shade_tables<C>(sc), sc.bsdfs[bsdf], st.rng, q);
    else
        q.wi = st.wo, q.pdf = 1.0f, q.attenuation = V3{1.0f, 1.0f, 1.0f}, q.valid = true; // pass-through surface (quirk Q8)
    phase_mark(kPhaseBsdf);
    if (!q.valid)
    {
        end_sample();
        return;
    }
    st.wi = q.wi;
    st.pdf_sample = q.pdf;
    st.throughput *= q.attenuation / q.pdf;
    st.origin = surf.position;
    if (max_component(st.throughput) < kEps)
    {
        end_sample();
        return;
    }
    st.dir = -st.wi;
}
#endif // MCPT_WAVE_CODE

// Convenience for CPU-side emulation and unit tests: a whole pixel.
template <class C>
MCPT_HD V3 render_pixel(const DeviceScene &sc, uint32_t pixel, LaneCounters *cnt, bool independent = false, uint32_t seed = 0)
{
    PathState st;
    uint32_t stack[kWalkStackMax * kWalkStackStride];
    st.stack = stack;
    start_pixel(st, pixel);
    while (!pixel_done(sc, st))
    {
        if (!st.alive)
        {
            start_sample(sc, st, 1, independent, seed);
            if (cnt)
                ++cnt->samples;
        }
        path_step<C>(sc, st, cnt);
    }
    return pixel_value(sc, st);
}

} // namespace mcpt

#endif // MCPT_PATH_CORE_H
