// The QUEUED formulation of the path tracer: path slots, ray queues and per-material shade queues in HBM, one launch per
// stage.  Third sibling of path_core.h (a lane owns a path) and stream_core.h (a workgroup-local ray pool); it runs the
// same per-vertex functions (stream_vertex = path_step from "resolve" on, statement for statement) and consumes every
// pixel's random stream in the reference's order, so its frames are the reference's bit for bit.
//
// What is different from the first multi-kernel version (hip/wavefront_kernels.hip, kept for comparison):
//   * a slot is one compact RECORD (array of structures, 96 B: rng, pixel, sample, flags, depth, pdf, direction,
//     throughput, radiance, pixel sum [, origin, wo, medium for volume paths]) — a wavefront that shades 64 ARBITRARY
//     slots touches one or two 128-byte lines per slot, not one line per field (what made the material sort of round 2
//     slower than no sort: 52 field-major words per slot);
//   * slots are a POOL: a slot whose pixel is finished takes the next unassigned pixel (RenderJob::work_counter);
//   * the trace launch hands every answered extension ray to the shade queue OF THE MATERIAL IT HIT (wavefront ballot +
//     prefix count, one atomic per wavefront and queue), together with the hit record — the next round's shade launches
//     are one per BSDF kind, each compiled with that kind's model only (bsdfs.h, kOnly) and fed 64 slots of that kind
//     per wavefront: the divergent `switch` of the reference's megakernel (src/renderer/bsdfs/bsdf.cpp:188-211 inside
//     src/renderer/renderer.cpp:88-95) becomes a choice of queue;
//   * nothing waits for a shadow ray: the direct-light estimate of a vertex travels WITH its shadow ray
//     (contribution = throughput x estimate, computed at the vertex exactly as stream_fold would), and the lane that
//     finds the ray unoccluded adds it to the slot's radiance — the same float addition, done before anything else
//     touches that radiance, as `L += throughput * direct` is in connect_lights (path.cpp:138-236).
// One shadow ray per vertex (scenes with one emitter or area lights only: the mesh configurations); other scenes run the
// stream kernel.  Camera rays come from the primary-visibility pre-pass (hip/primary_kernel.hip): a sample starts at its
// first vertex.
//
// Everything here is MCPT_HD: tests/emu runs the same functions on the host, queue after queue, against the goldens.
#ifndef MCPT_QUEUE_CORE_H
#define MCPT_QUEUE_CORE_H

#include "stream_core.h"

namespace mcpt
{

// Kernel groups: which shade launch handles a vertex.  Group 0 has no BSDF code: extension rays that left the scene,
// emitters seen by a ray, surfaces without a BSDF (pass-through, quirk Q8); group g >= 1 is BsdfKind g + 1.
constexpr uint32_t kQueueGroups = 7;
MCPT_HD constexpr uint32_t queue_group_of_kind(uint32_t kind) { return kind >= kBsdfDiffuse && kind <= kBsdfPlastic ? kind - 1u : 0u; }
MCPT_HD constexpr uint32_t queue_kind_of_group(uint32_t group) { return group == 0 ? kBsdfNoCode : group + 1u; }
MCPT_HD uint32_t queue_group_of_instance(const DeviceScene &sc, uint32_t inst)
{
    const uint32_t b = sc.instances[inst].bsdf;
    return b == kNone ? 0u : queue_group_of_kind(sc.bsdfs[b].kind);
}

// Samples a slot may START in one shade call (a sample whose camera ray leaves the scene ends where it starts, and the
// next one starts in the same call): bounds what one lane does while the rest of its wavefront waits.
#ifndef MCPT_QUEUE_SAMPLES_PER_CALL
#define MCPT_QUEUE_SAMPLES_PER_CALL 16
#endif
constexpr uint32_t kQueueSamplesPerCall = MCPT_QUEUE_SAMPLES_PER_CALL;

// ---- slot record ------------------------------------------------------------------------------
constexpr uint32_t kQueueSlotWords = 24; // 96 B = 6 x 16 B
enum QueueSlotWord : uint32_t
{
    kQRng = 0, kQPixel, kQSample, kQFlags, // flags: StreamFlag bits 0..15 | medium << 16
    kQDepth = 4, kQPdf, kQDir,             // direction of the ray in flight (3)
    kQThroughput = 9, kQL = 12, kQSum = 15,
    kQOrigin = 18, kQWo = 21               // volume paths only
};
constexpr uint32_t kSlotRequeue = 1u << 13; // the slot goes straight to another round's shade queue (no ray out)
constexpr uint32_t kSlotFresh = 1u << 14;   // no pixel yet

// ---- queue records ----------------------------------------------------------------------------
// extension ray: origin, direction, slot, group of the shade queue the slot joins IF THE RAY MISSES        8 words
//                (the launch that starts the slot's next sample: queue_group_after_end)
// shadow ray:    origin, direction, t_max, id, contribution (3), -                  12 words
//                id = slot | kQueuePush (the slot has no extension ray out: this lane queues it for shading) |
//                     group << kQueueGroupShift (... in that group's queue)
// shade entry:   slot, primitive (kNone: miss, kQueueNoHit: nothing was traced), instance | inside << 31, a, b, c, t, -
constexpr uint32_t kQueueExtWords = 8, kQueueShadowWords = 12, kQueueEntryWords = 8;
constexpr uint32_t kQueueNoHit = 0xFFFFFFFEu;
constexpr uint32_t kQueuePush = 1u << 31, kQueueGroupShift = 28, kQueueSlotMask = (1u << kQueueGroupShift) - 1u;

struct QueueRays // what a shade call emits
{
    bool ext, shadow, requeue;
    uint32_t requeue_group;  // requeue: the group whose launch shades the slot's pending first vertex
    uint32_t shadow_id_bits; // kQueuePush | group << kQueueGroupShift, or 0
    uint32_t miss_group;     // ext: the shade queue the slot joins if its extension ray leaves the scene
    V3 contribution;         // shadow: what the slot's radiance gains if the ray is unoccluded
};

template <class C>
MCPT_HD void queue_load(const uint32_t *rec, StreamSlot<1> &s)
{
    PathState &st = s.st;
    st.rng = rec[kQRng], st.pixel = rec[kQPixel], st.sample = rec[kQSample];
    const uint32_t fm = rec[kQFlags];
    s.flags = fm & 0xFFFFu;
    st.medium = (fm >> 16) == 0xFFFFu ? kNone : (fm >> 16);
    st.depth = rec[kQDepth], st.pdf_sample = as_float(rec[kQPdf]);
    st.dir = V3{as_float(rec[kQDir]), as_float(rec[kQDir + 1]), as_float(rec[kQDir + 2])};
    st.throughput = V3{as_float(rec[kQThroughput]), as_float(rec[kQThroughput + 1]), as_float(rec[kQThroughput + 2])};
    st.L = V3{as_float(rec[kQL]), as_float(rec[kQL + 1]), as_float(rec[kQL + 2])};
    st.pixel_sum = V3{as_float(rec[kQSum]), as_float(rec[kQSum + 1]), as_float(rec[kQSum + 2])};
    st.origin = V3{0, 0, 0}, st.wo = V3{0, 0, 0}, st.wi = V3{0, 0, 0};
    if (C::kVolPath)
    {
        st.origin = V3{as_float(rec[kQOrigin]), as_float(rec[kQOrigin + 1]), as_float(rec[kQOrigin + 2])};
        st.wo = V3{as_float(rec[kQWo]), as_float(rec[kQWo + 1]), as_float(rec[kQWo + 2])};
    }
    s.item = 0;
    s.thr_connect = V3{0, 0, 0}, s.c[0] = V3{0, 0, 0};
    s.hit_valid = false, s.hit_t = kMaxFloat, s.occluded[0] = false;
    s.hit.prim = kNone, s.hit.inst = 0, s.hit.a = s.hit.b = s.hit.c = 0.0f, s.hit.inside = false;
}

template <class C>
MCPT_HD void queue_save(uint32_t *rec, const StreamSlot<1> &s)
{
    const PathState &st = s.st;
    rec[kQRng] = st.rng, rec[kQPixel] = st.pixel, rec[kQSample] = st.sample;
    rec[kQFlags] = (s.flags & 0xFFFFu) | ((st.medium == kNone ? 0xFFFFu : st.medium) << 16);
    rec[kQDepth] = st.depth, rec[kQPdf] = as_uint(st.pdf_sample);
    rec[kQDir] = as_uint(st.dir.x), rec[kQDir + 1] = as_uint(st.dir.y), rec[kQDir + 2] = as_uint(st.dir.z);
    rec[kQThroughput] = as_uint(st.throughput.x), rec[kQThroughput + 1] = as_uint(st.throughput.y), rec[kQThroughput + 2] = as_uint(st.throughput.z);
    rec[kQL] = as_uint(st.L.x), rec[kQL + 1] = as_uint(st.L.y), rec[kQL + 2] = as_uint(st.L.z);
    rec[kQSum] = as_uint(st.pixel_sum.x), rec[kQSum + 1] = as_uint(st.pixel_sum.y), rec[kQSum + 2] = as_uint(st.pixel_sum.z);
    if (C::kVolPath)
    {
        rec[kQOrigin] = as_uint(st.origin.x), rec[kQOrigin + 1] = as_uint(st.origin.y), rec[kQOrigin + 2] = as_uint(st.origin.z);
        rec[kQWo] = as_uint(st.wo.x), rec[kQWo + 1] = as_uint(st.wo.y), rec[kQWo + 2] = as_uint(st.wo.z);
    }
}

// The traced hit of a shade-queue entry -> the slot (kQueueNoHit entries carry none).
MCPT_HD void queue_take_entry(const uint32_t *entry, StreamSlot<1> &s)
{
    const uint32_t prim = entry[1];
    if (prim == kQueueNoHit)
        return;
    s.hit_valid = prim != kNone;
    if (s.hit_valid)
    {
        s.hit.prim = prim, s.hit.inst = entry[2] & 0x7fffffffu, s.hit.inside = (entry[2] >> 31) != 0;
        s.hit.a = as_float(entry[3]), s.hit.b = as_float(entry[4]), s.hit.c = as_float(entry[5]);
        s.hit_t = as_float(entry[6]);
    }
}

// The camera ray of the sample the slot has just started: its hit from the pre-pass (path_core.h::path_step does the same).
template <class C>
MCPT_HD void queue_primary_hit(const DeviceScene &sc, StreamSlot<1> &s)
{
    PathState &st = s.st;
    const uint32_t *rec = prehit_record(sc, st.pixel, st.sample - 1u); // (start_sample advanced it)
    const uint32_t prim = rec[0];
    s.hit_valid = prim != kNone, s.hit_t = kMaxFloat;
    if (s.hit_valid)
    {
        Ray ray = make_ray(st.origin, st.dir);
        hit_from_record<C::kAnalytic>(sc, rec[1], prim, ray, s.hit);
        s.hit_t = ray.t_max;
    }
}

// Which group's launch shades the slot after a path that ENDED with its shadow ray still out: that launch finishes the
// sample and starts the next one, whose first vertex is known from the pre-pass.
template <class C>
MCPT_HD uint32_t queue_group_after_end(const DeviceScene &sc, const PathState &st)
{
    if (st.sample >= sc.camera.spp)
        return 0u; // the pixel is finished: any launch can store it and take the next one
    const uint32_t *rec = prehit_record(sc, st.pixel, st.sample);
    return rec[0] != kNone ? queue_group_of_instance(sc, rec[1]) : 0u;
}

enum QueueShadeResult : uint32_t
{
    kQueueContinue = 0,  // the slot's rays / its requeue for this round are in `out` (or it is exhausted)
    kQueuePixelDone = 1, // all samples of the slot's pixel are finished: the caller stores pixel_value(), hands the slot
                         // its next pixel (start_pixel) or marks it exhausted, and calls again
};

// One slot, one round, in the launch of group kGroup.  `budget`: samples the slot may still start in this round.
// stream_shade (stream_core.h) with three differences: the vertex is shaded only if it belongs to this launch's group
// (otherwise the slot is requeued for that group's launch, its camera-ray hit being recomputed there from the pre-pass);
// the direct-light estimate leaves with the shadow ray; at most `budget` samples start.
template <class C, uint32_t kGroup>
MCPT_HD QueueShadeResult queue_shade(const DeviceScene &sc, StreamSlot<1> &s, uint32_t &budget, QueueRays &out, LaneCounters *cnt)
{
    constexpr uint32_t kOnly = queue_kind_of_group(kGroup);
    PathState &st = s.st;
    out.ext = out.shadow = out.requeue = false, out.requeue_group = 0, out.shadow_id_bits = 0, out.contribution = V3{0, 0, 0};
    out.miss_group = 0;
    stream_unpack(s);
    if (s.flags & kSlotExhausted)
        return kQueueContinue;
    s.flags &= ~(kSlotExtRay | kSlotShadow0 | kSlotRequeue | kSlotFresh);
    bool at_vertex = false;
    if (s.flags & kSlotEnded)
    {
        s.flags &= ~kSlotEnded;
        finish_sample(st); // (its shadow ray's lane has added the direct light meanwhile)
    }
    else if (st.alive)
    {
        if (st.primary)
        {
            // requeued here with its first vertex pending
            st.origin = from(sc.camera.eye);
            queue_primary_hit<C>(sc, s);
        }
        at_vertex = true;
    }
    for (;;)
    {
        if (at_vertex)
        {
            stream_vertex<C, 1, kOnly>(sc, s, cnt);
            at_vertex = false;
            if (s.flags & kSlotShadow0)
            {
                // stream_fold's addend, formed now: the sum over the vertex's lights starts from zero (connect_lights)
                out.shadow = true;
                out.contribution = s.thr_connect * (V3{0, 0, 0} + s.c[0]);
            }
            s.flags &= ~(kSlotPending | kSlotHasC0);
        }
        if (st.alive || (s.flags & kSlotEnded))
            break;
        if (st.sample >= sc.camera.spp)
        {
            stream_pack(s);
            return kQueuePixelDone;
        }
        if (budget == 0)
        {
            out.requeue = true, out.requeue_group = kGroup; // goes on next round, same launch
            break;
        }
        --budget;
        start_sample(sc, st);
        if (cnt)
            ++cnt->samples;
        queue_primary_hit<C>(sc, s);
        // (a camera ray that leaves the scene is shaded by whoever started it: every launch has that code)
        const uint32_t g = s.hit_valid ? queue_group_of_instance(sc, s.hit.inst) : kGroup;
        if (g != kGroup)
        {
            out.requeue = true, out.requeue_group = g;
            break;
        }
        at_vertex = true;
    }
    out.ext = st.alive && !out.requeue;
    if (out.requeue)
        s.flags |= kSlotRequeue;
    if (out.ext)
    {
        s.flags |= kSlotExtRay;
        // a ray that leaves the scene ends the sample (every launch has that code: stream_vertex's miss branch); the
        // launch that shades the NEXT sample's first vertex should be the one that does it, or the slot pays a round
        // for the hand-over
        out.miss_group = queue_group_after_end<C>(sc, st);
    }
    if (out.shadow && !out.ext) // ended at a vertex whose shadow ray is out: that ray's lane queues the slot
        out.shadow_id_bits = kQueuePush | (queue_group_after_end<C>(sc, st) << kQueueGroupShift);
    stream_pack(s);
    return kQueueContinue;
}

} // namespace mcpt

#endif // MCPT_QUEUE_CORE_H
