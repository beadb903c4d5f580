// The path tracer as a STREAM of rays through a workgroup-local pool (the production formulation for
// scenes without opacity masks), next to the lane-owns-a-path state machine of path_core.h.
//
// What it cures.  In path_core.h a lane owns a pixel and walks its own ray; a wavefront's walk lasts as
// long as its slowest lane's (cornell: a ray needs ~10 node visits, the wavefront runs ~34 steps: 29 % of
// the lanes do useful work in the node phase, 25 % in the primitive phase; 12-18 % on meshes).  Here a
// workgroup owns P path SLOTS (P = 2..4 x its lanes) whose state lives in memory, not in a lane's
// registers, and works in rounds of two phases separated by workgroup barriers:
//
//   shade   one slot per lane at a time: fold in the shadow-ray results of the previous step, finish /
//           regenerate samples, turn the hit of the slot's extension ray into the next vertex (surface
//           frame, emission, roulette, light sampling, BSDF sample).  A vertex EMITS up to 1 + S rays: the
//           next extension ray and S shadow rays (one per emitter + one for the sampled area light).  The
//           reference draws every random number of a vertex before it knows whether a shadow ray is
//           blocked (path.cpp:144-205: the draws precede each shadow query, and nothing after a query
//           draws), so the direct-light contribution is computed on the assumption "unoccluded", stored,
//           and added when the shadow ray's answer arrives — the LCG stream is consumed in the reference's
//           order and the image does not change.  Emitted rays are COMPACTED into the workgroup's ray list
//           with a wavefront ballot + prefix count (one LDS atomic per wavefront and ray kind); slots can be
//           handed to lanes SORTED BY MATERIAL (stream kernel, kSort) so that a wavefront shades one BSDF.
//   trace   the lanes drain the ray list: a lane whose ray is finished retires it (writes the hit / the
//           occlusion bit to the ray's slot) and FETCHES THE NEXT RAY from the list instead of idling until
//           the slowest lane of its wavefront is done; the wavefront picks, step by step, the phase (node
//           step or primitive test) that more of its lanes are waiting for.  A ray's own visiting order is
//           that of walk_ordered (traversal.h), so its answer is the same; which lane traces it is irrelevant.
//
// Register pressure drops as a side effect: no path state is live during a walk and no walk state during
// shading (the mesh instantiations of the lane-owns-a-path kernel spilled 8.9 GB of scratch per frame).
//
// Everything here is MCPT_HD: tests/emu runs the same functions on the host, slot after slot and ray after
// ray, and must reproduce the reference's golden frames bit for bit.
#ifndef MCPT_STREAM_CORE_H
#define MCPT_STREAM_CORE_H

#include "path_core.h"

namespace mcpt
{

constexpr uint32_t kStreamMaxShadow = 2; // shadow rays a vertex can emit in the stream formulation

enum StreamFlag : uint32_t
{
    kSlotAlive = 1u << 0,     // a path is in flight
    kSlotPrimary = 1u << 1,   // its ray in flight is the camera ray
    kSlotInMedium = 1u << 2,  // current vertex is a medium scattering event
    kSlotEnded = 1u << 3,     // the path ended at a vertex whose shadow rays are still out: finish after folding
    kSlotPending = 1u << 4,   // a direct-light estimate waits for its shadow rays
    kSlotHasC0 = 1u << 5,     // ... light j contributes (if unoccluded); bits 5 + j
    kSlotExtRay = 1u << 8,    // this round the slot emits an extension ray
    kSlotShadow0 = 1u << 9,   // ... shadow ray j; bits 9 + j
    kSlotExhausted = 1u << 12 // no pixel left for this slot
};

// A path slot in registers (shade phase).  `S` = shadow rays per vertex the kernel supports.
template <uint32_t S>
struct StreamSlot
{
    PathState st;     // rng, pixel, sample, depth, medium, pdf_sample, origin, dir, wo, throughput, L, pixel_sum
    uint32_t flags;   // StreamFlag
    uint32_t item;    // work item (pixel in tile order) the slot is rendering
    V3 thr_connect;   // throughput at the vertex whose direct light is pending
    V3 c[S];          // direct-light estimate of light j at that vertex, if unoccluded
    // result of the extension ray (written by the trace phase)
    HitRaw hit;
    bool hit_valid;
    float hit_t;
    // shadow rays emitted at the current vertex; after the trace phase occluded[j]
    V3 sh_origin[S], sh_dir[S];
    float sh_tmax[S];
    bool occluded[S];
};

// ---- shade ------------------------------------------------------------------------------------
// Direct light at a vertex, deferred: draws the vertex's light-sampling numbers in the reference's
// order, writes one shadow ray per contributing light and its estimate c[j].  Mirrors connect_lights
// (path_core.h) statement by statement; what differs is only WHEN the occlusion answer is used.
template <class C, uint32_t S, uint32_t kOnly = 0>
MCPT_HD void stream_connect(const DeviceScene &sc, StreamSlot<S> &s, bool at_medium, const Surface &surf, V3 position)
{
    PathState &st = s.st;
    const LightTables LT = light_tables<C>(sc);
    const uint32_t bsdf = at_medium ? kNone : sc.instances[surf.inst].bsdf;
    const V3 wo = st.wo;
    uint32_t conn_medium = kNone;
    if (C::kVolPath)
        conn_medium = at_medium ? st.medium : (sc.integrator.volpath ? medium_on_side(sc, surf, true, wo) : kNone);
    const bool vol = C::kVolPath && sc.integrator.volpath != 0;

    auto weigh = [&](V3 wi, float distance, V3 &tr, V3 &att, float &pdf) -> bool
    {
        tr = V3{1.0f, 1.0f, 1.0f};
        if (at_medium)
        {
            MediumEvent m = medium_event_init();
            m.distance = distance;
            medium_transmittance(sc.media[st.medium], m);
            if (!m.valid)
                return false;
            tr = m.attenuation / m.pdf;
            PhaseQuery p;
            p.wi = wi, p.wo = wo;
            phase_eval(sc.media[st.medium], p);
            if (!p.valid)
                return false;
            att = p.attenuation, pdf = p.pdf;
            return true;
        }
        if (dot(-wi, surf.normal) < kEpsFloat)
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
        const BsdfQuery q = eval_at<C, kOnly>(sc, surf, bsdf, wi, wo);
        if (!q.valid)
            return false;
        att = q.attenuation, pdf = q.pdf;
        return true;
    };

    uint32_t j = 0; // next shadow-ray record
    auto emit = [&](V3 origin, V3 dir, float t_max, V3 estimate)
    {
        // (the launcher guarantees n_emitters + (area lights ? 1 : 0) <= S)
#pragma unroll
        for (uint32_t k = 0; k < S; ++k)
            if (k == j)
            {
                s.sh_origin[k] = origin, s.sh_dir[k] = dir, s.sh_tmax[k] = t_max, s.c[k] = estimate;
                s.flags |= (kSlotHasC0 << k) | (kSlotShadow0 << k);
            }
    };

    if (C::kEmitters)
    {
        for (uint32_t k = 0; k < sc.integrator.n_emitters; ++k, ++j)
        {
            const EmitterRec &e = sc.emitters[k];
            const float xi0 = lcg_next(st.rng), xi1 = lcg_next(st.rng);
            const LightSample ls = emitter_sample(LT, e, position, xi0, xi1);
            V3 tr, att;
            float pdf;
            if (!weigh(ls.wi, ls.distance, tr, att, pdf))
                continue;
            const V3 radiance = emitter_eval_sample(LT, e, ls);
            if (ls.harsh)
            {
                emit(position, -ls.wi, ls.distance - kEpsDistance, vol ? radiance * tr * att : radiance * att);
            }
            else
            {
                const float pdf_direct = emitter_pdf(LT, e, -ls.wi);
                if (pdf_direct > kEpsFloat)
                {
                    const float w = power_heuristic(pdf_direct, pdf);
                    emit(position, -ls.wi, ls.distance - kEpsDistance,
                         vol ? w * radiance * tr * att / pdf_direct : w * radiance * (att / pdf_direct));
                }
            }
        }
    }

    if (sc.integrator.n_area_lights != 0)
    {
        const float xi_pick = lcg_next(st.rng);
        const uint32_t light = cdf_search(sc.integrator.n_area_lights + 1, sc.light_cdf, xi_pick) - 1;
        const uint32_t inst = sc.light_inst[light];
        const float xi0 = lcg_next(st.rng), xi1 = lcg_next(st.rng), xi2 = lcg_next(st.rng);
        const LightPoint lp = sample_instance<C::kAnalytic>(sc, inst, xi0, xi1, xi2);
        const V3 d = position - lp.position;
        const float distance = length(d);
        const V3 wi = normalize(d);
        const float cos_light = dot(wi, lp.normal);
        if (cos_light < kEpsFloat)
            return;
        V3 tr, att;
        float pdf;
        if (!weigh(wi, distance, tr, att, pdf))
            return;
        const float pdf_direct = area_light_pdf(sc, light, inst, distance, cos_light), w = power_heuristic(pdf_direct, pdf);
        const V3 radiance = texture_color(sc.textures, sc.texels, sc.bsdfs[sc.instances[inst].bsdf].tex0, lp.uv, !C::kTextures);
        // the shadow ray starts ON THE LIGHT and travels to the shading point (path.cpp:207-214)
        emit(lp.position, wi, distance - kEpsDistance, vol ? w * (radiance * tr * att / pdf_direct) : w * radiance * (att / pdf_direct));
    }
}

// The occlusion answers of the previous vertex's shadow rays are in: L += throughput * (sum of the
// unoccluded estimates), the sum built in light order from zero as connect_lights builds it.
template <uint32_t S>
MCPT_HD void stream_fold(StreamSlot<S> &s)
{
    if (!(s.flags & kSlotPending))
        return;
    V3 direct = V3{0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < S; ++k)
        if ((s.flags & (kSlotHasC0 << k)) && !s.occluded[k])
            direct += s.c[k];
    s.st.L += s.thr_connect * direct;
    s.flags &= ~(kSlotPending | (kSlotHasC0 * ((1u << S) - 1u)));
}

// Brings the PathState booleans and the flag word in line (the flag word is what is stored).
template <uint32_t S>
MCPT_HD void stream_unpack(StreamSlot<S> &s)
{
    s.st.alive = (s.flags & kSlotAlive) != 0, s.st.primary = (s.flags & kSlotPrimary) != 0;
    s.st.in_medium = (s.flags & kSlotInMedium) != 0;
}
template <uint32_t S>
MCPT_HD void stream_pack(StreamSlot<S> &s)
{
    s.flags = (s.flags & ~(kSlotAlive | kSlotPrimary | kSlotInMedium)) | (s.st.alive ? kSlotAlive : 0u) |
              (s.st.primary ? kSlotPrimary : 0u) | (s.st.in_medium ? kSlotInMedium : 0u);
}

// The vertex at the end of the slot's extension ray: path_step (path_core.h) from "resolve" on, with the
// shadow queries deferred.  On return the slot is alive with its next extension ray in st.origin / st.dir
// (kSlotExtRay), or the sample is finished (st.alive == false), or it has ENDED but waits for shadow rays
// (kSlotEnded).
// kOnly (bsdfs.h): the BSDF kind the caller guarantees for a surface with a BSDF that is not an emitter, or 0.
template <class C, uint32_t S, uint32_t kOnly = 0>
MCPT_HD void stream_vertex(const DeviceScene &sc, StreamSlot<S> &s, LaneCounters *cnt)
{
    PathState &st = s.st;
    const IntegratorRec &ig = sc.integrator;
    const bool vol = C::kVolPath && ig.volpath != 0;
    const LightTables LT = light_tables<C>(sc);
    const bool hit_valid = s.hit_valid;
    Ray ray; // what the rest of the step reads of the ray that was traced
    ray.origin = st.origin, ray.dir = st.dir, ray.t_max = s.hit_t;
    st.wi = -st.dir; // (after a scatter wi is the sampled direction and the ray leaves along -wi)
    if (!C::kVolPath)
        st.wo = -st.dir; // primary: wo = -look; later vertices: wo = wi before its first use (path.cpp:127)

    Surface surf;
    if (hit_valid)
    {
        surf = make_surface<C::kAnalytic, C::kTextures>(sc, ray, s.hit);
        if (cnt)
            ++cnt->shaded_hits;
    }
    else
    {
        surf.inside = false, surf.inst = 0, surf.uv = V2{0, 0};
        surf.position = surf.normal = surf.tangent = surf.bitangent = V3{0, 0, 0};
    }

    // ---- resolve -------------------------------------------------------------
    if (st.primary && !hit_valid)
    {
        if (ig.id_envmap != kNone)
            st.L += emitter_eval_dir(LT, sc.emitters[ig.id_envmap], st.dir);
        if (ig.id_sun != kNone)
            st.L += emitter_eval_dir(LT, sc.emitters[ig.id_sun], st.dir);
        finish_sample(st);
        return;
    }

    if (vol)
    {
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

    const uint32_t bsdf = hit_valid ? sc.instances[surf.inst].bsdf : kNone;
    if (!st.in_medium)
    {
        if (!hit_valid)
        {
            if (ig.id_envmap != kNone)
            {
                const EmitterRec &env = sc.emitters[ig.id_envmap];
                const V3 radiance = emitter_eval_dir(LT, env, -st.wi);
                const float pdf_direct = emitter_pdf(LT, env, -st.wi), w = power_heuristic(st.pdf_sample, pdf_direct);
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
                finish_sample(st);
                return;
            }
            if (b.kind == kBsdfAreaLight)
            {
                const V3 radiance = texture_color(sc.textures, sc.texels, b.tex0, surf.uv, !C::kTextures);
                if (st.primary)
                {
                    if (!ig.hide_emitters)
                        st.L = radiance;
                }
                else
                {
                    const float cos_light = dot(st.wi, surf.normal);
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
            st.wo = st.wi;
            if (st.depth >= ig.depth_rr)
                st.throughput *= ig.rr_scale;
        }
    }

    // ---- roulette ------------------------------------------------------------
    st.depth = st.primary ? 1u : st.depth + 1u;
    st.primary = false;
    if (!(st.depth < ig.depth_rr || (st.depth < ig.depth_max && lcg_next(st.rng) < ig.pdf_rr)))
    {
        finish_sample(st);
        return;
    }

    // ---- connect (deferred) --------------------------------------------------
    const V3 vertex = st.in_medium ? st.origin : surf.position;
    s.thr_connect = st.throughput;
    s.flags |= kSlotPending;
    stream_connect<C, S, kOnly>(sc, s, st.in_medium, surf, vertex);
    const bool shadows_out = (s.flags & (kSlotShadow0 * ((1u << S) - 1u))) != 0;
    if (!shadows_out)
        stream_fold(s); // nothing to wait for: L += throughput * 0, as the reference does

    // ---- scatter -------------------------------------------------------------
    bool ended = false;
    if (vol && st.in_medium)
    {
        PhaseQuery p;
        p.wo = st.wo;
        phase_sample(sc.media[st.medium], st.rng, p);
        if (!p.valid)
            ended = true;
        else
        {
            st.wi = p.wi;
            st.throughput *= p.attenuation / p.pdf;
            st.pdf_sample = p.pdf;
        }
    }
    else
    {
        BsdfQuery q = query_at(surf, st.wo, st.wo);
        if (bsdf != kNone)
            bsdf_sample<C::kMicrofacet, kOnly, C::kKinds>(shade_tables<C>(sc), sc.bsdfs[bsdf], st.rng, q);
        else
            q.wi = st.wo, q.pdf = 1.0f, q.attenuation = V3{1.0f, 1.0f, 1.0f}, q.valid = true;
        if (!q.valid)
            ended = true;
        else
        {
            st.wi = q.wi;
            st.pdf_sample = q.pdf;
            st.throughput *= q.attenuation / q.pdf;
            st.origin = surf.position;
        }
    }
    if (!ended && max_component(st.throughput) < kEps)
        ended = true;
    if (ended)
    {
        if (shadows_out)
            s.flags |= kSlotEnded, st.alive = false; // finish_sample once the shadow answers are folded in
        else
            finish_sample(st);
        return;
    }
    st.dir = -st.wi;
}

enum StreamShadeResult : uint32_t
{
    kStreamContinue = 0,  // the slot emitted its rays for this round (or is exhausted)
    kStreamPixelDone = 1, // all samples of the slot's pixel are finished: the caller stores pixel_value(), hands the
                          // slot its next pixel (start_pixel) or marks it exhausted, and calls again
};

// One slot, one round.  Call again after handling kStreamPixelDone.
// With the primary-visibility pre-pass (sc.prehit) a new sample's camera ray is not emitted: its hit is known, the
// sample's first vertex is shaded at once — and if the sample ends there (a camera ray that leaves the scene, an
// emitter seen directly, ...) the next sample starts in the same call.
// `independent` / `seed`: the independent-sample RNG mode (mcpt_renderer_set_rng) without split samples — every sample
// starts its own PCG-hashed stream (path_core.h::start_sample).
// `plane_items` (split samples, step > 1): the slot's item is k * plane_items + the pixel's work item.
template <class C, uint32_t S>
MCPT_HD StreamShadeResult stream_shade(const DeviceScene &sc, StreamSlot<S> &s, LaneCounters *cnt, bool independent = false,
                                       uint32_t seed = 0, uint32_t step = 1, uint32_t plane_items = 0)
{
    PathState &st = s.st;
    stream_unpack(s);
    if (s.flags & kSlotExhausted)
        return kStreamContinue;
    stream_fold(s);
    s.flags &= ~(kSlotExtRay | (kSlotShadow0 * ((1u << S) - 1u)));
    bool at_vertex = false; // a hit record waits to be shaded: the traced extension ray's, or a camera ray's from the pre-pass
    if (s.flags & kSlotEnded)
    {
        s.flags &= ~kSlotEnded;
        finish_sample(st);
    }
    else
        at_vertex = st.alive;
    for (;;) // (ONE call site of stream_vertex: the shading code is inlined once)
    {
        if (at_vertex)
        {
            stream_vertex<C, S>(sc, s, cnt); // alive with its next ray, ended (waiting for shadow answers), or finished
            at_vertex = false;
        }
        if (st.alive || (s.flags & kSlotEnded))
            break;
        if (st.sample >= sc.camera.spp)
        {
            stream_pack(s);
            return kStreamPixelDone;
        }
        start_sample(sc, st, step, independent, seed);
        if (cnt)
            ++cnt->samples;
        if (sc.prehit == nullptr)
            break;
        // (the slot knows its work item: no pixel -> item arithmetic per sample)
        const uint32_t item = step == 1 ? s.item : s.item % plane_items;
        const uint32_t *rec = sc.prehit + 2 * (static_cast<size_t>(item) * sc.camera.spp + (st.sample - step));
        const uint32_t prim = rec[0];
        s.hit_valid = prim != kNone, s.hit_t = kMaxFloat;
        if (s.hit_valid)
        {
            Ray ray = make_ray(st.origin, st.dir);
            hit_from_record<C::kAnalytic>(sc, rec[1], prim, ray, s.hit);
            s.hit_t = ray.t_max;
        }
        at_vertex = true;
    }
    if (st.alive)
        s.flags |= kSlotExtRay;
    stream_pack(s);
    return kStreamContinue;
}

// Primitive test of the pooled walk: test_slot (traversal.h) with the query kind as a run-time value, so
// that extension and shadow rays can share a wavefront.  The expensive part (the watertight probe) is
// common to both kinds.
template <bool kAnalytic, bool kSlivers, bool kLeafCheck = false>
MCPT_HD bool pool_test_slot(const DeviceScene &sc, uint32_t slot, bool any, Ray &ray, HitRaw &hit, ClosestState &best)
{
    if (any)
        return test_slot<true, kAnalytic, kSlivers, kLeafCheck>(sc, slot, ray, hit, best);
    return test_slot<false, kAnalytic, kSlivers, kLeafCheck>(sc, slot, ray, hit, best);
}

// Experiment switch (-DMCPT_STREAM_WIDE=1 on a unit whose instantiations all run outside LDS): the trace phase walks the
// 4-wide quantised hierarchy with the short stack (short_stack.h) instead of the binary one.
#ifndef MCPT_STREAM_WIDE
#define MCPT_STREAM_WIDE 0
#endif
// SPECULATIVE SEARCH (-DMCPT_STREAM_SPECULATE=1): a lane whose walk arrives at a primitive does not wait for the wavefront's
// primitive phase — it sets the primitive aside (one per lane) and goes on searching from its stack with the bound it has; the
// primitive is tested in the next primitive phase.  The answers do not change: a later test can only have left the bound
// larger than it would have been (more nodes visited, never fewer), and the primitive test decides ties by replaying the
// reference on the pair, whatever the order (traversal.h, test_slot).
// When the wavefront leaves its node phase for a primitive phase: holding lanes x NUM > searching lanes x DEN.  Round 3, one box,
// 1:1 / 2:1 / 1:2: matpreview rough conductor 209.5 / 209.9 / 216.1 ms, rough dielectric 317.2 / 316.3 / 328.2, dragon 160.8 / 162.6 / 169.6.
#ifndef MCPT_STREAM_HOLD_NUM
#define MCPT_STREAM_HOLD_NUM 1u
#endif
#ifndef MCPT_STREAM_HOLD_DEN
#define MCPT_STREAM_HOLD_DEN 1u
#endif
#ifndef MCPT_STREAM_SPECULATE
#define MCPT_STREAM_SPECULATE 0
#endif

// ---- slot storage -----------------------------------------------------------------------------
// Structure of arrays over the P slots of a workgroup: field f of slot i at [f * P + i], so that the
// lanes of a wavefront (consecutive slots) read consecutive words.  "hot" = what the trace phase touches
// (rays in, hits out): LDS for scenes whose traversal stacks leave room, otherwise global memory; "cold" =
// the rest of the path state, touched once per round by the slot's shading lane: global memory (the
// workgroup's own region, cache resident).
struct StreamStore
{
    uint32_t *hot, *cold;
    uint32_t P;
};

// hot fields.  The extension ray's origin and its hit record share three words: the tracing lane reads
// the origin when it fetches the ray and overwrites it with the barycentrics when it retires it.
enum StreamHot : uint32_t
{
    kHotPrim = 0,  // out: primitive (kNone: miss)
    kHotInst = 1,  // out: instance | inside << 31
    kHotA = 2,     // in: origin.x   out: a
    kHotB = 3,     // in: origin.y   out: b
    kHotC = 4,     // in: origin.z   out: c
    kHotT = 5,     // out: distance
    kHotDir = 6,   // in: direction (3 words, kept)
    kHotShadow = 9 // per shadow ray j, 7 words at kHotShadow + 7 j: origin, direction, t_max (out: < 0 = occluded)
};
MCPT_HD constexpr uint32_t stream_hot_words(uint32_t S) { return kHotShadow + 7u * S; }

enum StreamCold : uint32_t
{
    kColdRng = 0, kColdPixel, kColdItem, kColdSample, kColdDepth, kColdFlags, kColdPdf,
    kColdThroughput, kColdL = kColdThroughput + 3, kColdSum = kColdL + 3, kColdThrConnect = kColdSum + 3,
    kColdDir = kColdThrConnect + 3, kColdMedium = kColdDir + 3, kColdWo, kColdEstimates = kColdWo + 3 // 3 words per shadow ray
};
MCPT_HD constexpr uint32_t stream_cold_words(uint32_t S) { return kColdEstimates + 3u * S + 3u; } // + the ray origin (volpath)

MCPT_HD V3 stream_get3(const uint32_t *base, uint32_t P, uint32_t field, uint32_t i)
{
    return V3{as_float(base[field * P + i]), as_float(base[(field + 1) * P + i]), as_float(base[(field + 2) * P + i])};
}
MCPT_HD void stream_put3(uint32_t *base, uint32_t P, uint32_t field, uint32_t i, V3 v)
{
    base[field * P + i] = as_uint(v.x), base[(field + 1) * P + i] = as_uint(v.y), base[(field + 2) * P + i] = as_uint(v.z);
}

// The answers of the trace phase -> registers (the slot's flags say which rays it had out).
template <class C, uint32_t S>
MCPT_HD void stream_load_hot(const StreamStore &m, uint32_t i, StreamSlot<S> &s)
{
    const uint32_t P = m.P;
    s.hit_valid = false, s.hit_t = kMaxFloat;
    s.hit.prim = kNone, s.hit.inst = 0, s.hit.a = s.hit.b = s.hit.c = 0.0f, s.hit.inside = false;
    if (s.flags & kSlotExtRay)
    {
        const uint32_t prim = m.hot[kHotPrim * P + i];
        if (prim != kNone)
        {
            const uint32_t inst = m.hot[kHotInst * P + i];
            s.hit_valid = true, s.hit.prim = prim, s.hit.inst = inst & 0x7fffffffu, s.hit.inside = (inst >> 31) != 0;
            s.hit.a = as_float(m.hot[kHotA * P + i]), s.hit.b = as_float(m.hot[kHotB * P + i]);
            s.hit.c = as_float(m.hot[kHotC * P + i]), s.hit_t = as_float(m.hot[kHotT * P + i]);
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < S; ++k)
        s.occluded[k] = (s.flags & (kSlotShadow0 << k)) != 0 && as_float(m.hot[(kHotShadow + 7u * k + 6u) * P + i]) < 0.0f;
}

// The rays the slot emits this round -> the pool.
template <class C, uint32_t S>
MCPT_HD void stream_save_hot(const StreamStore &m, uint32_t i, const StreamSlot<S> &s)
{
    const uint32_t P = m.P;
    if (s.flags & kSlotExtRay)
        stream_put3(m.hot, P, kHotA, i, s.st.origin), stream_put3(m.hot, P, kHotDir, i, s.st.dir);
#pragma unroll
    for (uint32_t k = 0; k < S; ++k)
        if (s.flags & (kSlotShadow0 << k))
        {
            stream_put3(m.hot, P, kHotShadow + 7u * k, i, s.sh_origin[k]);
            stream_put3(m.hot, P, kHotShadow + 7u * k + 3u, i, s.sh_dir[k]);
            m.hot[(kHotShadow + 7u * k + 6u) * P + i] = as_uint(s.sh_tmax[k]);
        }
}

// Path state of a slot that lives in memory between rounds (the variants with more slots than lanes):
// memory -> registers at the start of the slot's shade step ...
template <class C, uint32_t S>
MCPT_HD void stream_load(const StreamStore &m, uint32_t i, StreamSlot<S> &s)
{
    const uint32_t P = m.P;
    PathState &st = s.st;
    s.flags = m.cold[kColdFlags * P + i];
    st.rng = m.cold[kColdRng * P + i], st.pixel = m.cold[kColdPixel * P + i], s.item = m.cold[kColdItem * P + i];
    st.sample = m.cold[kColdSample * P + i], st.depth = m.cold[kColdDepth * P + i];
    st.pdf_sample = as_float(m.cold[kColdPdf * P + i]);
    st.throughput = stream_get3(m.cold, P, kColdThroughput, i), st.L = stream_get3(m.cold, P, kColdL, i);
    st.pixel_sum = stream_get3(m.cold, P, kColdSum, i);
    st.medium = kNone, st.wo = V3{0, 0, 0}, st.wi = V3{0, 0, 0};
    st.origin = V3{0, 0, 0};
    if (C::kVolPath)
    {
        st.medium = m.cold[kColdMedium * P + i], st.wo = stream_get3(m.cold, P, kColdWo, i);
        // (the tracing lane overwrites the ray's origin with the hit record; the free-flight sampling of a
        //  medium reads it after the walk)
        st.origin = stream_get3(m.cold, P, kColdEstimates + 3u * S, i);
    }
    st.dir = stream_get3(m.cold, P, kColdDir, i);
    s.thr_connect = stream_get3(m.cold, P, kColdThrConnect, i);
#pragma unroll
    for (uint32_t k = 0; k < S; ++k)
        s.c[k] = stream_get3(m.cold, P, kColdEstimates + 3u * k, i);
    stream_load_hot<C, S>(m, i, s);
}

// ... and back at its end, including the rays it emits this round.
template <class C, uint32_t S>
MCPT_HD void stream_save(const StreamStore &m, uint32_t i, const StreamSlot<S> &s)
{
    const uint32_t P = m.P;
    const PathState &st = s.st;
    m.cold[kColdFlags * P + i] = s.flags;
    m.cold[kColdRng * P + i] = st.rng, m.cold[kColdPixel * P + i] = st.pixel, m.cold[kColdItem * P + i] = s.item;
    m.cold[kColdSample * P + i] = st.sample, m.cold[kColdDepth * P + i] = st.depth;
    m.cold[kColdPdf * P + i] = as_uint(st.pdf_sample);
    stream_put3(m.cold, P, kColdThroughput, i, st.throughput), stream_put3(m.cold, P, kColdL, i, st.L);
    stream_put3(m.cold, P, kColdSum, i, st.pixel_sum);
    if (C::kVolPath)
    {
        m.cold[kColdMedium * P + i] = st.medium, stream_put3(m.cold, P, kColdWo, i, st.wo);
        stream_put3(m.cold, P, kColdEstimates + 3u * S, i, st.origin);
    }
    stream_put3(m.cold, P, kColdDir, i, st.dir);
    stream_put3(m.cold, P, kColdThrConnect, i, s.thr_connect);
#pragma unroll
    for (uint32_t k = 0; k < S; ++k)
        stream_put3(m.cold, P, kColdEstimates + 3u * k, i, s.c[k]);
    stream_save_hot<C, S>(m, i, s);
}

// ---- trace ------------------------------------------------------------------------------------
// Wavefront helpers (a "wavefront" of the host build is one lane).
// The ray list of a round: extension rays (closest queries, the long ones) are listed from the front,
// shadow rays behind them, so that the pool is drained longest-first.
struct StreamRayList
{
    const uint32_t *ids; // [0, n_ext): extension rays; [P, P + n_shadow): shadow rays.  id = kind * P + slot
    uint32_t n_ext, n_shadow;
    uint32_t *next; // fetch cursor
};

// Drains the ray list.  Every lane of the workgroup calls this; lanes take rays from the list as they become
// free.  `refill_at`: a wavefront fetches when at least this many of its lanes are free (fetching costs the
// whole wavefront a few dozen instructions, so not for every single lane).
//
// `own` (may be null): the calling lane's own extension ray, straight from its registers — the variant with one
// slot per lane.  The lane starts with it (no fetch, no trip through the pool; every lane of the wavefront sets
// its ray up at the same time) and takes rays from the list — then only shadow rays — once it is done: the
// shadow rays of the workgroup fill the lanes that would otherwise wait for their wavefront's longest
// extension ray.  The answer comes back in `own`.
struct OwnRay
{
    bool active;   // in: the lane has an extension ray this round
    V3 origin, dir;
    bool found;    // out
    HitRaw hit;
    float t;
};
constexpr uint32_t kOwnRayId = 0xFFFFFFFEu;

// kListHasExt = false: the list holds shadow rays only (one slot per lane: every extension ray is its lane's own) — the
// retire / fetch code of listed extension rays is not compiled in.
template <class C, bool kCount, bool kListHasExt = true>
MCPT_HD void stream_trace(const DeviceScene &sc, const StreamStore &m, const StreamRayList &list, uint32_t *stack,
                          uint32_t refill_at, LaneCounters *cnt, OwnRay *own = nullptr)
{
    if (sc.integrator.n_walk_nodes == 0)
        return; // (an empty scene emits no rays that could hit anything; the launcher does not pick this kernel)
    const uint32_t P = m.P, total = list.n_ext + list.n_shadow;
    uint32_t my = kNone, slot = 0, kind = 0; // the ray this lane is tracing
    Ray ray = make_ray(V3{0, 0, 0}, V3{0, 0, 1});
    HitRaw hit;
    hit.inst = hit.prim = 0, hit.a = hit.b = hit.c = 0.0f, hit.inside = false;
    ClosestState best{false, 0.0f, 0};
    uint32_t depth = 1, cur = kWalkDone;
    constexpr bool kWideTrace = MCPT_STREAM_WIDE != 0;
    constexpr bool kSpeculate = MCPT_STREAM_SPECULATE != 0 && !kWideTrace;
    uint32_t held = kWalkDone; // kSpeculate: the primitive set aside (kWalkDone: none)
    ShortStack<kWideRing> wstack = wide_stack_of(sc, stack);
    auto begin_walk = [&]()
    {
        held = kWalkDone;
        if (kWideTrace)
            wstack.reset(), wstack.store(0, kWalkDone);
        else
            stack[0] = kWalkDone;
        depth = 1, cur = 0;
    };
    bool pool_empty = total == 0; // wavefront-uniform
    if (own)
    {
        own->found = false, own->t = kMaxFloat;
        if (own->active)
        {
            my = kOwnRayId, kind = 0;
            ray = make_ray(own->origin, own->dir);
            best = ClosestState{false, kMaxFloat, 0};
            begin_walk();
            if (cnt)
                ++cnt->closest_rays;
        }
    }
    // A lane is SEARCHING (cur is a node), HOLDING (cur is a primitive slot) or FREE (cur == kWalkDone: its ray is
    // finished and waits to be retired, or it has none).
    const uint32_t n_lanes = lanes_where(true);
    // the first fetch of a round does not wait for `refill_at` free lanes: lanes without an own ray take listed rays at
    // once, so that a round's shadow rays run NEXT TO its extension rays, not behind them (a frame can be as long as
    // its longest pixel's chain of rounds: dragon/scene.xml)
    uint32_t fetch_at = own ? 1u : refill_at;
    for (;;)
    {
        // ---- node phase: a tight loop of node steps (walk_ordered, traversal.h) that runs while the lanes that
        //      hold a primitive are not more than the searching ones and too few lanes are free to fetch ----
        for (;;)
        {
            const bool searching = !(cur & kWalkLeaf);
            const uint32_t n_searching = lanes_where(searching);
            if (n_searching == 0)
                break;
            const uint32_t n_free = lanes_where(cur == kWalkDone && (!kSpeculate || held == kWalkDone)), n_holding = n_lanes - n_searching - n_free;
            if (MCPT_STREAM_HOLD_NUM * n_holding > MCPT_STREAM_HOLD_DEN * n_searching || (!pool_empty && n_free >= fetch_at))
                break; // (each exit is followed by progress below: a primitive phase or a fetch)
            if (searching && kWideTrace)
            {
                if (kCount)
                {
                    cnt->node_tests += 2;
                    if (is_leading_lane())
                        ++cnt->wave_node_steps;
                }
                wide_node_step(sc, wstack, ray, cur, depth);
            }
            else if (searching)
            {
                const float4 *n = sc.walk_nodes + 4 * static_cast<size_t>(cur);
                const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
                if (kCount)
                {
                    cnt->node_tests += 2;
                    if (is_leading_lane())
                        ++cnt->wave_node_steps;
                }
                float enter0, enter1;
                const bool hit0 = box_enter(n0, n1, ray, enter0), hit1 = box_enter(n2, n3, ray, enter1);
                const uint32_t ref0 = as_uint(n0.w), ref1 = as_uint(n1.w);
                const bool first0 = enter0 <= enter1, both = hit0 && hit1, none = !(hit0 || hit1);
                const uint32_t toward = (hit0 && (first0 || !hit1)) ? ref0 : ref1;
                const uint32_t postponed = stack[(depth - 1) * kWalkStackStride];
                stack[depth * kWalkStackStride] = first0 ? ref1 : ref0;
                depth = depth + (both ? 1u : 0u) - (none ? 1u : 0u);
                cur = none ? postponed : toward;
                if (kSpeculate)
                {
                    // arrived at a primitive with nothing set aside yet: set it aside and take the next postponed reference
                    // (depth >= 1 here: entry 0 is the sentinel)
                    const bool aside = (cur & kWalkLeaf) != 0 && cur != kWalkDone && held == kWalkDone;
                    const uint32_t below = stack[(depth ? depth - 1u : 0u) * kWalkStackStride];
                    held = aside ? cur : held;
                    depth -= aside ? 1u : 0u;
                    cur = aside ? below : cur;
                }
            }
        }
        // ---- primitive phase: every lane that holds a primitive tests it ----
        // (kSpeculate: the primitive set aside comes first and the lane's cursor stays where it is; a lane that is still
        //  searching tests its set-aside primitive too — the wavefront is in its primitive phase anyway, and the bound shrinks
        //  earlier)
        const bool from_aside = kSpeculate && held != kWalkDone;
        if (from_aside || ((cur & kWalkLeaf) != 0 && cur != kWalkDone))
        {
            if (kCount)
            {
                ++cnt->prim_tests;
                if (is_leading_lane())
                    ++cnt->wave_prim_steps;
            }
            const bool any = kind != 0;
            const uint32_t p = from_aside ? held : cur;
            held = kWalkDone;
            if (pool_test_slot<C::kAnalytic, C::kSlivers, kWideTrace>(sc, p & ~kWalkLeaf, any, ray, hit, best) && any)
                cur = kWalkDone;
            else if (!from_aside)
            {
                --depth;
                cur = kWideTrace ? wstack.load(depth) : stack[depth * kWalkStackStride];
            }
        }
        // ---- retire finished rays, fetch new ones ----
        const bool lane_free = cur == kWalkDone && (!kSpeculate || held == kWalkDone);
        const uint32_t n_free = lanes_where(lane_free);
        if (n_free == n_lanes || (n_free >= fetch_at && !pool_empty))
        {
            fetch_at = refill_at;
            if (my != kNone && lane_free)
            {
                if (my == kOwnRayId)
                {
                    own->found = best.found, own->hit = hit, own->t = best.found ? best.best_t : kMaxFloat;
                }
                else if (kListHasExt && kind == 0)
                {
                    m.hot[kHotPrim * P + slot] = best.found ? hit.prim : kNone;
                    if (best.found)
                    {
                        m.hot[kHotInst * P + slot] = hit.inst | (hit.inside ? 0x80000000u : 0u);
                        m.hot[kHotA * P + slot] = as_uint(hit.a), m.hot[kHotB * P + slot] = as_uint(hit.b);
                        m.hot[kHotC * P + slot] = as_uint(hit.c), m.hot[kHotT * P + slot] = as_uint(best.best_t);
                    }
                }
                else if (best.found)
                    m.hot[(kHotShadow + 7u * (kind - 1u) + 6u) * P + slot] = as_uint(-1.0f);
                my = kNone;
            }
            if (pool_empty)
            {
                if (n_free == n_lanes)
                    break;
                continue;
            }
            const bool want = my == kNone;
            const uint32_t k = wave_reserve(list.next, want);
            if (want && k < total)
            {
                my = kListHasExt ? list.ids[k < list.n_ext ? k : P + (k - list.n_ext)] : list.ids[P + k];
                kind = kListHasExt ? (my >= P ? 1u : 0u) + (my >= 2u * P ? 1u : 0u) : (my >= 2u * P ? 2u : 1u), slot = my - kind * P;
                V3 o, d;
                float t_max = kMaxFloat;
                if (kListHasExt && kind == 0)
                    o = stream_get3(m.hot, P, kHotA, slot), d = stream_get3(m.hot, P, kHotDir, slot);
                else
                {
                    const uint32_t f = kHotShadow + 7u * (kind - 1u);
                    o = stream_get3(m.hot, P, f, slot), d = stream_get3(m.hot, P, f + 3u, slot);
                    t_max = as_float(m.hot[(f + 6u) * P + slot]);
                }
                ray = make_ray(o, d);
                ray.t_max = t_max;
                best = ClosestState{false, t_max, 0};
                begin_walk();
                if (cnt)
                {
                    if (kind == 0)
                        ++cnt->closest_rays;
                    else
                        ++cnt->shadow_rays;
                }
            }
            // the indices handed out grow with the lane rank: once one of them reaches the end of the list, the
            // list is drained (other wavefronts may still be tracing what they fetched)
            pool_empty = lanes_where(want && k + 1u >= total) != 0;
        }
    }
}

} // namespace mcpt

#endif // MCPT_STREAM_CORE_H
