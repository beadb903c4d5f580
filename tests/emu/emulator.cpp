// TEST INFRASTRUCTURE — host emulation of the GPU lane state machine.
//
// Compiles the product's kernel body (csrc/path_core.h and everything it
// includes) and the product's scene commit (csrc/host/commit.cpp) for the CPU
// and runs one "lane" per pixel in a thread pool.  It exists so that the
// logic of the HIP kernel (stackless traversal, stage ordering, draw order,
// expression forms) can be checked bit-for-bit against the oracle in a
// container without a GPU.  It is NOT part of libmcpt_hip.so, is never used by
// the product path, bench.py or smoke(), and is not a fallback: the product has
// no CPU rendering path.
#include <array>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "host/commit.hpp"
#include "path_core.h"
#include "stream_core.h"
#include "queue_core.h"
#include "short_stack.h"

namespace
{

using namespace mcpt;

thread_local std::string g_error;
// the throughput random modes of mcpt_renderer_set_rng in RenderAll: independent stream per (pixel, sample) — PCG-hashed in
// libmcpt_emu.so, Owen-scrambled Sobol points in libmcpt_emu_ld.so (the same source compiled with MCPT_LOW_DISCREPANCY)
bool g_independent = false;
uint32_t g_rng_seed = 0;

template <uint32_t kFeatures>
void RenderAll(const DeviceScene &sc, float *frame, LaneCounters *total)
{
    using C = Config<kFeatures>;
    const uint32_t n = static_cast<uint32_t>(sc.camera.width) * sc.camera.height;
    const unsigned workers = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<uint32_t> next{0};
    std::vector<LaneCounters> counts(workers, LaneCounters{});
    auto work = [&](unsigned tid)
    {
        for (;;)
        {
            const uint32_t begin = next.fetch_add(64);
            if (begin >= n)
                break;
            for (uint32_t p = begin; p < std::min(begin + 64, n); ++p)
            {
                LaneCounters c{};
                const V3 v = render_pixel<C>(sc, p, total ? &c : nullptr, g_independent, g_rng_seed);
                frame[3 * p] = v.x, frame[3 * p + 1] = v.y, frame[3 * p + 2] = v.z;
                counts[tid].closest_rays += c.closest_rays, counts[tid].shadow_rays += c.shadow_rays;
                counts[tid].node_tests += c.node_tests, counts[tid].prim_tests += c.prim_tests;
                counts[tid].shaded_hits += c.shaded_hits, counts[tid].samples += c.samples;
            }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < workers; ++t)
        pool.emplace_back(work, t);
    work(0);
    for (std::thread &t : pool)
        t.join();
    if (total)
        for (const LaneCounters &c : counts)
        {
            total->closest_rays += c.closest_rays, total->shadow_rays += c.shadow_rays;
            total->node_tests += c.node_tests, total->prim_tests += c.prim_tests;
            total->shaded_hits += c.shaded_hits, total->samples += c.samples;
        }
}

// The stream formulation (stream_core.h) on the host: "workgroups" of P slots run their rounds
// (shade every slot, list the emitted rays, trace the list) one after the other.  Same functions as the
// HIP stream kernel; only the lane loop and the ray-list bookkeeping are restated here.
template <uint32_t kFeatures, uint32_t S>
void RenderAllStream(const DeviceScene &sc, float *frame, LaneCounters *total, uint32_t P)
{
    using C = Config<kFeatures>;
    const uint32_t n = static_cast<uint32_t>(sc.camera.width) * sc.camera.height;
    const uint32_t n_blocks = (n + P - 1) / P;
    const unsigned workers = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<uint32_t> next_block{0};
    std::mutex mu;
    auto work = [&]()
    {
        std::vector<uint32_t> hot(stream_hot_words(S) * P), cold(stream_cold_words(S) * P), ids((1 + S) * P);
        std::vector<uint32_t> stack(kWalkStackMax * kWalkStackStride);
        LaneCounters cnt{};
        for (;;)
        {
            const uint32_t b = next_block.fetch_add(1);
            if (b >= n_blocks)
                break;
            const StreamStore m{hot.data(), cold.data(), P};
            std::fill(cold.begin(), cold.end(), 0u);
            // slots: one pixel each (the frame is cut into n_blocks runs of P pixels)
            for (uint32_t i = 0; i < P; ++i)
            {
                StreamSlot<S> s{};
                const uint32_t pixel = b * P + i;
                s.flags = pixel < n ? 0u : kSlotExhausted;
                s.item = pixel;
                if (pixel < n)
                    start_pixel(s.st, pixel);
                stream_save<C, S>(m, i, s);
            }
            for (;;)
            {
                uint32_t n_ext = 0, n_shadow = 0;
                for (uint32_t i = 0; i < P; ++i)
                {
                    StreamSlot<S> s;
                    stream_load<C, S>(m, i, s);
                    while (stream_shade<C, S>(sc, s, &cnt) == kStreamPixelDone)
                    {
                        const V3 v = pixel_value(sc, s.st);
                        frame[3 * s.st.pixel] = v.x, frame[3 * s.st.pixel + 1] = v.y, frame[3 * s.st.pixel + 2] = v.z;
                        s.flags |= kSlotExhausted;
                    }
                    stream_save<C, S>(m, i, s);
                    if (s.flags & kSlotExtRay)
                        ids[n_ext++] = i;
                    for (uint32_t k = 0; k < S; ++k)
                        if (s.flags & (kSlotShadow0 << k))
                            ids[P + n_shadow++] = (1 + k) * P + i;
                }
                if (n_ext + n_shadow == 0)
                    break;
                uint32_t cursor = 0;
                const StreamRayList list{ids.data(), n_ext, n_shadow, &cursor};
                stream_trace<C, true>(sc, m, list, stack.data(), 1, &cnt);
            }
        }
        if (total)
        {
            std::lock_guard<std::mutex> lock(mu);
            total->closest_rays += cnt.closest_rays, total->shadow_rays += cnt.shadow_rays;
            total->node_tests += cnt.node_tests, total->prim_tests += cnt.prim_tests;
            total->shaded_hits += cnt.shaded_hits, total->samples += cnt.samples;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < workers; ++t)
        pool.emplace_back(work);
    work();
    for (std::thread &t : pool)
        t.join();
}

// The QUEUED formulation (queue_core.h) on the host: one pool of slots, rounds of (trace every queued ray; shade every
// material group's queue).  Same functions as the HIP kernels (hip/queued_kernels.*): queue_shade, queue_load / queue_save,
// queue_take_entry, the contribution that travels with the shadow ray; only the queue bookkeeping is restated.  The
// camera-ray pre-pass is computed first, like on the GPU.  Single-threaded rounds (the point is the logic, not speed).
template <uint32_t kFeatures>
void RenderAllQueued(DeviceScene sc, float *frame, uint32_t n_slots, uint32_t *rounds_out)
{
    using C = Config<kFeatures>;
    const uint32_t w = static_cast<uint32_t>(sc.camera.width), h = static_cast<uint32_t>(sc.camera.height), spp = sc.camera.spp;
    const uint32_t tiles_x = (w + 7) / 8, tiles_y = (h + 7) / 8, n_items = tiles_x * tiles_y * 64;
    std::vector<uint32_t> stack(kWalkStackMax * kWalkStackStride);
    // pre-pass: the closest hit of every camera ray
    std::vector<uint32_t> prehit(size_t(2) * n_items * spp);
    {
        const unsigned workers = std::max(1u, std::thread::hardware_concurrency());
        std::atomic<uint32_t> next{0};
        auto work = [&]()
        {
            std::vector<uint32_t> st(kWalkStackMax * kWalkStackStride);
            for (;;)
            {
                const uint32_t p = next.fetch_add(1);
                if (p >= w * h)
                    break;
                for (uint32_t k = 0; k < spp; ++k)
                {
                    PathState ps;
                    ps.pixel = p, ps.sample = k;
                    start_sample(sc, ps);
                    Ray ray = make_ray(ps.origin, ps.dir);
                    HitRaw hit;
                    TraceStats ts{0, 0, 0, 0};
                    const bool found = walk_ordered<false, C::kAnalytic, false, C::kSlivers>(sc, st.data(), ray, hit, ts);
                    const size_t at = 2 * (size_t(item_of_pixel(p, w, tiles_x, 0, 1)) * spp + k); // (by work item, like hip/primary_kernel.hip)
                    prehit[at] = found ? hit.prim : kNone, prehit[at + 1] = found ? hit.inst : 0u;
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < workers; ++t)
            pool.emplace_back(work);
        work();
        for (std::thread &t : pool)
            t.join();
    }
    sc.prehit = prehit.data(), sc.prehit_step = 1;
    sc.prehit_tile_first = 0, sc.prehit_tile_stride = 1, sc.prehit_tiles_x = tiles_x;
    n_slots = std::max(1u, std::min(n_slots ? n_slots : n_items, n_items));
    std::vector<uint32_t> slots(size_t(n_slots) * kQueueSlotWords, 0u);
    struct Ext { V3 o, d; uint32_t id, miss_group; };
    // the trace launch's short stack (short_stack.h) with a ring of TWO entries: nearly every walk spills and reloads
    std::vector<uint32_t> ring(2 * kWalkStackStride), spill(kWalkStackMax);
    ShortStack<2> short_stack{ring.data(), spill.data(), 1u, 0u};
    struct Shadow { V3 o, d; float t_max; uint32_t id; V3 contribution; };
    struct Entry { uint32_t w[kQueueEntryWords]; };
    std::vector<Ext> ext, ext_next;
    std::vector<Shadow> shadow, shadow_next;
    std::vector<Entry> queue[2][kQueueGroups];
    uint32_t work_counter = 0;
    auto assign = [&](StreamSlot<1> &s, uint32_t item)
    {
        for (;; item = n_slots + work_counter++)
        {
            if (item >= n_items)
            {
                s.flags = kSlotExhausted;
                return;
            }
            const uint32_t tile = item >> 6, r = item & 63u;
            const uint32_t x = (tile % tiles_x) * 8u + (r & 7u), y = (tile / tiles_x) * 8u + (r >> 3);
            if (x < w && y < h)
            {
                s.flags = 0;
                start_pixel(s.st, y * w + x);
                return;
            }
        }
    };
    // one slot in the launch of group g: what the lane of queued_shade does
    auto shade_one = [&](auto group_tag, uint32_t id, const Entry *entry, uint32_t parity)
    {
        constexpr uint32_t kGroup = decltype(group_tag)::value;
        StreamSlot<1> s{};
        uint32_t *rec = slots.data() + size_t(id) * kQueueSlotWords;
        if (!entry)
        {
            s.st.medium = kNone;
            assign(s, id);
        }
        else
        {
            queue_load<C>(rec, s);
            queue_take_entry(entry->w, s);
        }
        QueueRays rays{};
        uint32_t budget = kQueueSamplesPerCall;
        while (!(s.flags & kSlotExhausted) && queue_shade<C, kGroup>(sc, s, budget, rays, nullptr) == kQueuePixelDone)
        {
            const V3 v = pixel_value(sc, s.st);
            frame[3 * size_t(s.st.pixel)] = v.x, frame[3 * size_t(s.st.pixel) + 1] = v.y, frame[3 * size_t(s.st.pixel) + 2] = v.z;
            assign(s, n_slots + work_counter++);
        }
        if (s.flags & kSlotExhausted)
            return;
        queue_save<C>(rec, s);
        if (rays.ext)
            ext_next.push_back(Ext{s.st.origin, s.st.dir, id, rays.miss_group});
        if (rays.shadow)
            shadow_next.push_back(Shadow{s.sh_origin[0], s.sh_dir[0], s.sh_tmax[0], id | rays.shadow_id_bits, rays.contribution});
        if (rays.requeue)
        {
            Entry e{};
            e.w[0] = id, e.w[1] = kQueueNoHit;
            queue[parity ^ 1u][rays.requeue_group].push_back(e);
        }
    };
    auto shade_group = [&](uint32_t g, uint32_t id, const Entry *entry, uint32_t parity)
    {
        switch (g)
        {
        case 0: shade_one(std::integral_constant<uint32_t, 0>{}, id, entry, parity); break;
        case 1: shade_one(std::integral_constant<uint32_t, 1>{}, id, entry, parity); break;
        case 2: shade_one(std::integral_constant<uint32_t, 2>{}, id, entry, parity); break;
        case 3: shade_one(std::integral_constant<uint32_t, 3>{}, id, entry, parity); break;
        case 4: shade_one(std::integral_constant<uint32_t, 4>{}, id, entry, parity); break;
        case 5: shade_one(std::integral_constant<uint32_t, 5>{}, id, entry, parity); break;
        default: shade_one(std::integral_constant<uint32_t, 6>{}, id, entry, parity); break;
        }
    };
    uint32_t round = 0;
    for (uint32_t id = 0; id < n_slots; ++id) // round 0
        shade_group(0, id, nullptr, 0);
    for (round = 1;; ++round)
    {
        const uint32_t parity = round & 1u;
        ext.swap(ext_next), shadow.swap(shadow_next);
        ext_next.clear(), shadow_next.clear();
        for (uint32_t g = 0; g < kQueueGroups; ++g)
            queue[parity ^ 1u][g].clear(); // (what queued_trace zeroes)
        size_t pending = ext.size() + shadow.size();
        for (uint32_t g = 0; g < kQueueGroups; ++g)
            pending += queue[parity][g].size();
        if (pending == 0)
            break;
        // ---- trace ----
        for (const Ext &r : ext)
        {
            Ray ray = make_ray(r.o, r.d);
            HitRaw hit;
            TraceStats ts{0, 0, 0, 0};
            (void)ts;
            const bool found = walk_ordered_short<false, C::kAnalytic, C::kSlivers, 2>(sc, short_stack, ray, hit);
            Entry e{};
            e.w[0] = r.id, e.w[1] = kNone;
            uint32_t group = r.miss_group;
            if (found)
            {
                group = queue_group_of_instance(sc, hit.inst);
                e.w[1] = hit.prim, e.w[2] = hit.inst | (hit.inside ? 0x80000000u : 0u), e.w[3] = as_uint(hit.a), e.w[4] = as_uint(hit.b);
                e.w[5] = as_uint(hit.c), e.w[6] = as_uint(ray.t_max);
            }
            queue[parity][group].push_back(e);
        }
        for (const Shadow &r : shadow)
        {
            Ray ray = make_ray(r.o, r.d);
            ray.t_max = r.t_max;
            HitRaw hit;
            TraceStats ts{0, 0, 0, 0};
            (void)ts;
            if (!walk_ordered_short<true, C::kAnalytic, C::kSlivers, 2>(sc, short_stack, ray, hit))
            {
                uint32_t *L = slots.data() + size_t(r.id & kQueueSlotMask) * kQueueSlotWords + kQL;
                L[0] = as_uint(as_float(L[0]) + r.contribution.x), L[1] = as_uint(as_float(L[1]) + r.contribution.y);
                L[2] = as_uint(as_float(L[2]) + r.contribution.z);
            }
            if (r.id & kQueuePush)
            {
                Entry e{};
                e.w[0] = r.id & kQueueSlotMask, e.w[1] = kQueueNoHit;
                queue[parity][(r.id >> kQueueGroupShift) & 7u].push_back(e);
            }
        }
        // ---- shade: one "launch" per group ----
        for (uint32_t g = 0; g < kQueueGroups; ++g)
            for (size_t k = 0; k < queue[parity][g].size(); ++k)
            {
                const Entry e = queue[parity][g][k];
                shade_group(g, e.w[0], &e, parity);
            }
    }
    if (rounds_out)
        *rounds_out = round;
}

} // namespace

extern "C"
{

// The queued formulation on the host (see RenderAllQueued).  Scenes the queued renderer accepts: surface paths on
// triangle meshes without opacity masks, at most one shadow ray per vertex.  n_slots: size of the slot pool (0 = one per
// work item).  rounds: number of rounds the frame took (may be null).
int mcpt_emu_render_queued(const char *mcsd_path, float *frame, uint32_t n_slots, uint32_t *rounds)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        const uint32_t n_shadow = flat.integrator.n_emitters + (flat.integrator.n_area_lights ? 1u : 0u);
        constexpr uint32_t kSurfaceF = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
        if (flat.integrator.has_masks || n_shadow > 1 || flat.integrator.n_walk_nodes == 0 || (flat.features & ~kSurfaceF) != 0)
            throw std::runtime_error("not a scene for the queued renderer (opacity masks, more than one light sample per vertex, volume paths, quadrics, or empty)");
        RenderAllQueued<kSurfaceF | kFeatOrderedWalk | kFeatSlivers>(sc, frame, n_slots, rounds);
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// The stream formulation on the host (see RenderAllStream).  The scene must be one the stream kernel
// accepts: no opacity masks, at most kStreamMaxShadow shadow rays per vertex.
int mcpt_emu_render_stream(const char *mcsd_path, float *frame, uint32_t slots_per_block, uint32_t *counters)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        const uint32_t n_shadow = flat.integrator.n_emitters + (flat.integrator.n_area_lights ? 1u : 0u);
        if (flat.integrator.has_masks || n_shadow > kStreamMaxShadow || flat.integrator.n_walk_nodes == 0)
            throw std::runtime_error("not a scene for the stream kernel (opacity masks, too many lights, or empty)");
        LaneCounters total{};
        constexpr uint32_t kAllF = kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet;
        constexpr uint32_t kSurfaceF = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
        constexpr uint32_t kO = kFeatOrderedWalk;
        const uint32_t f = flat.features;
        const bool slivers = flat.integrator.walk_sliver_reach > 0.0f;
        const uint32_t P = slots_per_block ? slots_per_block : 512;
        LaneCounters *cnt = &total;
        if (slivers)
        {
            if ((f & ~kSurfaceF) == 0)
                n_shadow <= 1 ? RenderAllStream<kSurfaceF | kO | kFeatSlivers, 1>(sc, frame, cnt, P)
                              : RenderAllStream<kSurfaceF | kO | kFeatSlivers, 2>(sc, frame, cnt, P);
            else
                n_shadow <= 1 ? RenderAllStream<kAllF | kO | kFeatSlivers, 1>(sc, frame, cnt, P)
                              : RenderAllStream<kAllF | kO | kFeatSlivers, 2>(sc, frame, cnt, P);
        }
        else if (f == 0)
            RenderAllStream<kO, 1>(sc, frame, cnt, P);
        else if ((f & ~kFeatEmitters) == 0)
            n_shadow <= 1 ? RenderAllStream<kFeatEmitters | kO, 1>(sc, frame, cnt, P)
                          : RenderAllStream<kFeatEmitters | kO, 2>(sc, frame, cnt, P);
        else if ((f & ~kSurfaceF) == 0)
            n_shadow <= 1 ? RenderAllStream<kSurfaceF | kO, 1>(sc, frame, cnt, P)
                          : RenderAllStream<kSurfaceF | kO, 2>(sc, frame, cnt, P);
        else
            n_shadow <= 1 ? RenderAllStream<kAllF | kO, 1>(sc, frame, cnt, P) : RenderAllStream<kAllF | kO, 2>(sc, frame, cnt, P);
        if (counters)
        {
            counters[0] = total.closest_rays, counters[1] = total.shadow_rays, counters[2] = total.node_tests;
            counters[3] = total.prim_tests, counters[4] = total.shaded_hits, counters[5] = total.samples;
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

const char *mcpt_emu_last_error(void) { return g_error.c_str(); }

// 0 = production split rule of the ordered-walk hierarchy, 1 = exact sweep, 2 = median,
// 3 = children swapped (commit.hpp).  Process-wide.
void mcpt_emu_set_walk_tree(int strategy) { SetWalkTreeStrategyForTesting(strategy); }

// mcpt_emu_render with one independent stream per (pixel, sample) (mcpt_renderer_set_rng modes 1 / 2)
void mcpt_emu_set_rng(int independent, uint32_t seed) { g_independent = independent != 0, g_rng_seed = seed; }
int mcpt_emu_low_discrepancy() { return MCPT_LOW_DISCREPANCY_ACTIVE; }
// n successive draws of the random stream from the state `word` (this build's generator: vecmath.h)
void mcpt_emu_draws(uint32_t word, uint32_t n, float *out)
{
    for (uint32_t k = 0; k < n; ++k)
        out[k] = lcg_next(word);
}
// the word a sample of the low-discrepancy build starts from
uint32_t mcpt_emu_ld_pack(uint32_t sample, uint32_t seed, uint32_t pixel) { return ld_pack(sample, pcg_hash(pcg_hash(seed) + pixel)); }

// variant: -1 = pick like the GPU launcher does, otherwise a feature mask to
// force (must be a superset of the scene's features).  Bit kFeatOrderedWalk of a
// forced mask selects the ordered walk; -1 uses it whenever the launcher would
// (no opacity masks), -2 = launcher's pick but with the reference-order walk.
// counters: 6 x u32 or NULL.
int mcpt_emu_render(const char *mcsd_path, float *frame, int variant, uint32_t *counters, uint32_t *features_out)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        if (features_out)
            *features_out = flat.features;
        LaneCounters total{};
        LaneCounters *cnt = counters ? &total : nullptr;
        constexpr uint32_t kAll = kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet;
        uint32_t f = flat.features;
        uint32_t pick;
        bool ordered = variant == -1 && flat.integrator.has_masks == 0;
        if (variant >= 0)
        {
            ordered = (static_cast<uint32_t>(variant) & kFeatOrderedWalk) != 0;
            pick = static_cast<uint32_t>(variant) & ~(kFeatOrderedWalk | kFeatWideWalk);
        }
        else if (f == 0)
            pick = 0;
        else if ((f & ~kFeatEmitters) == 0)
            pick = kFeatEmitters;
        else if ((f & ~(kFeatEmitters | kFeatTextures | kFeatMicrofacet)) == 0)
            pick = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
        else
            pick = kAll;
        if ((f & ~pick) != 0)
            throw std::runtime_error("forced variant does not cover the scene's features");
        if (ordered && flat.integrator.has_masks)
            throw std::runtime_error("the ordered walk cannot be used with opacity masks");
        // like the launcher: plain ordered walk in the two lean instantiations, vote-scheduled
        // (a no-op on the one-lane "wavefronts" of the host build) in the others
        constexpr uint32_t kO = kFeatOrderedWalk, kV = kFeatOrderedWalk | kFeatVoteWalk;
        constexpr uint32_t kSurfaceF = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
        if (ordered && variant >= 0 && (static_cast<uint32_t>(variant) & kFeatWideWalk))
        {
            // the production walk of scenes outside LDS: the 4-wide quantised hierarchy (full-feature instantiation)
            RenderAll<kAll | kV | kFeatSlivers | kFeatWideWalk>(sc, frame, cnt);
        }
        else if (ordered && flat.integrator.walk_sliver_reach > 0.0f)
        {
            // like the launcher: scenes with sliver triangles run the sliver-aware instantiations
            if ((pick & ~kSurfaceF) == 0)
                RenderAll<kSurfaceF | kV | kFeatSlivers>(sc, frame, cnt);
            else
                RenderAll<kAll | kV | kFeatSlivers>(sc, frame, cnt);
        }
        else
        switch (pick | (ordered ? kO : 0u))
        {
        case 0:
            RenderAll<0>(sc, frame, cnt);
            break;
        case kO:
            RenderAll<kO>(sc, frame, cnt);
            break;
        case kFeatEmitters:
            RenderAll<kFeatEmitters>(sc, frame, cnt);
            break;
        case kFeatEmitters | kO:
            RenderAll<kFeatEmitters | kO>(sc, frame, cnt);
            break;
        case kFeatEmitters | kFeatTextures | kFeatMicrofacet:
            RenderAll<kFeatEmitters | kFeatTextures | kFeatMicrofacet>(sc, frame, cnt);
            break;
        case kFeatEmitters | kFeatTextures | kFeatMicrofacet | kO:
            RenderAll<kFeatEmitters | kFeatTextures | kFeatMicrofacet | kV>(sc, frame, cnt);
            break;
        default:
            if (ordered)
                RenderAll<kAll | kV>(sc, frame, cnt);
            else
                RenderAll<kAll>(sc, frame, cnt);
            break;
        }
        if (counters)
        {
            counters[0] = total.closest_rays, counters[1] = total.shadow_rays, counters[2] = total.node_tests;
            counters[3] = total.prim_tests, counters[4] = total.shaded_hits, counters[5] = total.samples;
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// Wavefront model: runs the 64 pixels of each 8x8 tile in lock step, one path_step per
// lane per round like the GPU kernel, and accumulates what a wavefront pays for the
// walks under two schedules: "separate" = closest walk then shadow walk, each lasting
// as long as its slowest lane; "paired" = the shadow walk of a round runs in the same
// loop as the NEXT round's closest walk (a lane does one after the other), lasting as
// long as the largest per-lane sum.  out: {lane node steps, separate wave node steps,
// paired wave node steps, lane prim tests, separate wave prim phases, paired wave prim phases}.
int mcpt_emu_wave_model(const char *mcsd_path, double *out)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        if (flat.integrator.has_masks)
            throw std::runtime_error("masked scene");
        using C = Config<kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet | kFeatOrderedWalk | kFeatVoteWalk>;
        const uint32_t w = sc.camera.width, h = sc.camera.height;
        const uint32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
        std::vector<double> acc(6, 0.0);
        std::mutex mu;
        std::atomic<uint32_t> next{0};
        auto work = [&]()
        {
            std::vector<double> a(6, 0.0);
            for (;;)
            {
                const uint32_t tile = next.fetch_add(1);
                if (tile >= tx * ty)
                    break;
                PathState st[64];
                LaneCounters cnt[64];
                std::vector<uint32_t> stacks(64 * kWalkStackMax);
                bool has[64];
                for (uint32_t l = 0; l < 64; ++l)
                {
                    const uint32_t x = (tile % tx) * 8 + (l & 7), y = (tile / tx) * 8 + (l >> 3);
                    has[l] = x < w && y < h;
                    cnt[l] = LaneCounters{};
                    st[l].stack = &stacks[l * kWalkStackMax];
                    if (has[l])
                        start_pixel(st[l], y * w + x);
                }
                uint32_t carry_nodes[64] = {}, carry_prims[64] = {}; // previous round's shadow walk
                for (;;)
                {
                    bool any = false;
                    uint32_t cn[64] = {}, cp[64] = {}, sn[64] = {}, sp[64] = {};
                    for (uint32_t l = 0; l < 64; ++l)
                    {
                        if (!has[l])
                            continue;
                        if (!st[l].alive)
                        {
                            if (st[l].sample >= sc.camera.spp)
                            {
                                has[l] = false;
                                continue;
                            }
                            start_sample(sc, st[l]);
                        }
                        any = true;
                        path_step<C>(sc, st[l], &cnt[l]);
                        cn[l] = cnt[l].last_closest_nodes / 2, cp[l] = cnt[l].last_closest_prims;
                        sn[l] = cnt[l].last_shadow_nodes / 2, sp[l] = cnt[l].last_shadow_prims;
                    }
                    uint32_t mcn = 0, mcp = 0, msn = 0, msp = 0, mpn = 0, mpp = 0;
                    for (uint32_t l = 0; l < 64; ++l)
                    {
                        a[0] += cn[l] + sn[l], a[3] += cp[l] + sp[l];
                        mcn = std::max(mcn, cn[l]), mcp = std::max(mcp, cp[l]);
                        msn = std::max(msn, sn[l]), msp = std::max(msp, sp[l]);
                        mpn = std::max(mpn, cn[l] + carry_nodes[l]), mpp = std::max(mpp, cp[l] + carry_prims[l]);
                        carry_nodes[l] = sn[l], carry_prims[l] = sp[l];
                    }
                    a[1] += mcn + msn, a[4] += mcp + msp;
                    a[2] += mpn, a[5] += mpp;
                    if (!any)
                        break;
                }
            }
            std::lock_guard<std::mutex> lock(mu);
            for (int i = 0; i < 6; ++i)
                acc[i] += a[i];
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < std::max(1u, std::thread::hardware_concurrency()); ++t)
            pool.emplace_back(work);
        work();
        for (std::thread &t : pool)
            t.join();
        for (int i = 0; i < 6; ++i)
            out[i] = acc[i];
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// Pooled-traversal model: the rays of one wavefront phase (<= 64) share one LIFO pool of
// (ray, node) items; every step the top 64 items are processed, one per lane.  Returns
// steps and visits so that it can be compared with the per-lane walk.
struct PoolModel
{
    double node_steps = 0, node_visits = 0, prim_steps = 0, prim_tests = 0;
};

void RunPool(const DeviceScene &sc, const std::vector<std::array<float, 7>> &rays, bool any, PoolModel &m)
{
    struct Item
    {
        uint32_t ray, ref;
    };
    const size_t n = rays.size();
    if (n == 0 || sc.integrator.n_walk_nodes == 0)
        return;
    std::vector<Ray> r(n);
    std::vector<uint64_t> best(n, ~0ull);
    std::vector<char> done(n, 0);
    for (size_t i = 0; i < n; ++i)
    {
        r[i] = make_ray(V3{rays[i][0], rays[i][1], rays[i][2]}, V3{rays[i][3], rays[i][4], rays[i][5]});
        r[i].t_max = rays[i][6];
    }
    std::vector<Item> nodes, leaves;
    for (size_t i = 0; i < n; ++i)
        nodes.push_back(Item{static_cast<uint32_t>(i), 0u});
    while (!nodes.empty() || !leaves.empty())
    {
        // policy: run the primitive phase when a full wavefront of leaves waits or no nodes are left
        if (leaves.size() >= 64 || nodes.empty())
        {
            const size_t take = std::min<size_t>(64, leaves.size());
            std::vector<Item> batch(leaves.end() - take, leaves.end());
            leaves.resize(leaves.size() - take);
            m.prim_steps += 1;
            for (const Item &it : batch)
            {
                if (done[it.ray])
                    continue;
                m.prim_tests += 1;
                const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(it.ref & ~kWalkLeaf);
                Ray &ray = r[it.ray];
                const SlotHit sh = triangle_probe(p, ray);
                if (sh.hit && !(sh.t > ray.t_max))
                {
                    ray.t_max = sh.t;
                    if (any)
                        done[it.ray] = 1;
                }
            }
            continue;
        }
        const size_t take = std::min<size_t>(64, nodes.size());
        std::vector<Item> batch(nodes.end() - take, nodes.end());
        nodes.resize(nodes.size() - take);
        m.node_steps += 1;
        // lanes take items from the top downwards; their pushes land in lane order
        for (size_t l = 0; l < take; ++l)
        {
            const Item it = batch[take - 1 - l];
            if (done[it.ray])
                continue;
            m.node_visits += 1;
            const float4 *q = sc.walk_nodes + 4 * static_cast<size_t>(it.ref);
            float e0, e1;
            const bool h0 = box_enter(q[0], q[1], r[it.ray], e0), h1 = box_enter(q[2], q[3], r[it.ray], e1);
            const uint32_t r0 = as_uint(q[0].w), r1 = as_uint(q[1].w);
            const bool first0 = e0 <= e1;
            // far child first, so that the near one is on top
            const uint32_t order[2] = {first0 ? r1 : r0, first0 ? r0 : r1};
            const bool hit[2] = {first0 ? h1 : h0, first0 ? h0 : h1};
            for (int k = 0; k < 2; ++k)
                if (hit[k])
                    ((order[k] & kWalkLeaf) ? leaves : nodes).push_back(Item{it.ray, order[k]});
        }
    }
}

// Per-lane ordered walk of one wavefront phase, simulated in lock step, with
// SPECULATION: a lane that holds a primitive may keep searching (with its stale t_max)
// and collect up to `spec` more primitives while some lane that holds none is still
// searching; the primitive phase then tests every lane's collected primitives in the order
// found.  spec = 0 is the production while-while loop.  Returns wave-level node steps and
// primitive-phase iterations, and the lane-level visits / tests.
void RunSpeculative(const DeviceScene &sc, const std::vector<std::array<float, 7>> &rays, bool any, uint32_t spec,
                    PoolModel &m)
{
    const size_t n = rays.size();
    if (n == 0 || sc.integrator.n_walk_nodes == 0)
        return;
    struct Lane
    {
        Ray ray;
        std::vector<uint32_t> stack, pending;
        uint32_t cur = 0;
        bool done = false, found = false;
        uint32_t best_rank = 0;
    };
    std::vector<Lane> lanes(n);
    for (size_t i = 0; i < n; ++i)
    {
        lanes[i].ray = make_ray(V3{rays[i][0], rays[i][1], rays[i][2]}, V3{rays[i][3], rays[i][4], rays[i][5]});
        lanes[i].ray.t_max = rays[i][6];
    }
    auto pop = [](Lane &l)
    {
        if (l.stack.empty())
        {
            l.cur = kWalkDone;
            return;
        }
        l.cur = l.stack.back();
        l.stack.pop_back();
    };
    for (;;)
    {
        // ---- node phase ----
        for (;;)
        {
            bool needed = false; // a lane with nothing collected is still searching
            for (Lane &l : lanes)
                if (!l.done && l.pending.empty() && l.cur != kWalkDone && !(l.cur & kWalkLeaf))
                    needed = true;
            // lanes standing on a leaf collect it (free bookkeeping) and move on if allowed
            bool moved = true;
            while (moved)
            {
                moved = false;
                for (Lane &l : lanes)
                    if (!l.done && l.cur != kWalkDone && (l.cur & kWalkLeaf) && l.pending.size() < 1 + spec)
                    {
                        l.pending.push_back(l.cur);
                        pop(l);
                        moved = true;
                    }
            }
            needed = false;
            for (Lane &l : lanes)
                if (!l.done && l.pending.empty() && l.cur != kWalkDone)
                    needed = true;
            if (!needed)
                break;
            m.node_steps += 1;
            for (Lane &l : lanes)
            {
                if (l.done || l.cur == kWalkDone || (l.cur & kWalkLeaf))
                    continue; // finished, or holding a full set of primitives
                m.node_visits += 1;
                const float4 *q = sc.walk_nodes + 4 * static_cast<size_t>(l.cur);
                float e0, e1;
                const bool h0 = box_enter(q[0], q[1], l.ray, e0), h1 = box_enter(q[2], q[3], l.ray, e1);
                const uint32_t r0 = as_uint(q[0].w), r1 = as_uint(q[1].w);
                if (h0 && h1)
                {
                    const bool first0 = e0 <= e1;
                    l.stack.push_back(first0 ? r1 : r0);
                    l.cur = first0 ? r0 : r1;
                }
                else if (h0 || h1)
                    l.cur = h0 ? r0 : r1;
                else
                    pop(l);
            }
        }
        // ---- primitive phase: as many iterations as the fullest lane has collected ----
        size_t most = 0;
        bool anything = false;
        for (Lane &l : lanes)
        {
            most = std::max(most, l.done ? size_t(0) : l.pending.size());
            anything = anything || (!l.done && (!l.pending.empty() || l.cur != kWalkDone));
        }
        if (!anything)
            break;
        m.prim_steps += static_cast<double>(most);
        for (Lane &l : lanes)
        {
            if (l.done)
                continue;
            for (const uint32_t ref : l.pending)
            {
                m.prim_tests += 1;
                const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(ref & ~kWalkLeaf);
                const SlotHit sh = triangle_probe(p, l.ray);
                if (sh.hit && !(sh.t > l.ray.t_max))
                {
                    l.ray.t_max = sh.t;
                    if (any)
                    {
                        l.done = true;
                        break;
                    }
                }
            }
            l.pending.clear();
            if (!l.done && l.cur == kWalkDone)
                l.done = true;
        }
    }
}

// Pool model, second edition (round 4): closest and shadow rays of a round in ONE pool (`merged`), the primitive phase
// runs when `prim_at` leaves wait (or no node item is left), the culling bound of a closest ray is its best distance + the
// tie radius and is read at the START of a step (all lanes of a step see the same value, like 64 lanes reading LDS before
// any of them writes).  Records what a kernel has to be sized for: the longest item lists and the most hits one ray accepts.
struct Pool2Model
{
    double node_steps = 0, node_visits = 0, prim_steps = 0, prim_tests = 0, rounds = 0;
    double max_nodes = 0, max_leaves = 0, accepted = 0, rays = 0, accepted_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double node_lanes_hist[4] = {0, 0, 0, 0}; // node steps with 1-16, 17-32, 33-48, 49-64 lanes busy
};

void RunPool2(const DeviceScene &sc, const std::vector<std::array<float, 7>> &closest, const std::vector<std::array<float, 7>> &shadow,
              uint32_t prim_at, Pool2Model &m, bool wide = false, bool unordered = false)
{
    struct Item
    {
        uint32_t ray, ref;
    };
    const size_t nc = closest.size(), n = nc + shadow.size();
    if (n == 0 || sc.integrator.n_walk_nodes == 0)
        return;
    m.rounds += 1;
    const float tie = sc.integrator.walk_tie;
    std::vector<Ray> r(n);
    std::vector<float> bound(n);  // culling bound the NEXT step reads
    std::vector<uint32_t> accepted(n, 0);
    std::vector<char> done(n, 0);
    for (size_t i = 0; i < n; ++i)
    {
        const std::array<float, 7> &q = i < nc ? closest[i] : shadow[i - nc];
        r[i] = make_ray(V3{q[0], q[1], q[2]}, V3{q[3], q[4], q[5]});
        r[i].t_max = q[6], bound[i] = q[6];
    }
    std::vector<Item> nodes, leaves;
    for (size_t i = 0; i < n; ++i)
        nodes.push_back(Item{static_cast<uint32_t>(i), 0u});
    while (!nodes.empty() || !leaves.empty())
    {
        m.max_nodes = std::max<double>(m.max_nodes, nodes.size()), m.max_leaves = std::max<double>(m.max_leaves, leaves.size());
        if (leaves.size() >= prim_at || nodes.empty())
        {
            const size_t take = std::min<size_t>(64, leaves.size());
            std::vector<Item> batch(leaves.end() - take, leaves.end());
            leaves.resize(leaves.size() - take);
            m.prim_steps += 1;
            std::vector<float> next = bound;
            for (const Item &it : batch)
            {
                m.prim_tests += 1;
                const bool any = it.ray >= nc;
                if (any && done[it.ray])
                    continue;
                const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(it.ref & ~kWalkLeaf);
                const SlotHit sh = triangle_probe(p, r[it.ray]);
                if (sh.hit && !(sh.t > bound[it.ray]))
                {
                    ++accepted[it.ray];
                    if (any)
                        done[it.ray] = 1, next[it.ray] = -1.0f;
                    else
                        next[it.ray] = std::min(next[it.ray], sh.t + tie);
                }
            }
            bound = next;
            continue;
        }
        const size_t take = std::min<size_t>(64, nodes.size());
        std::vector<Item> batch(nodes.end() - take, nodes.end());
        nodes.resize(nodes.size() - take);
        m.node_steps += 1;
        m.node_lanes_hist[(take - 1) / 16] += 1;
        std::vector<Item> far_nodes, near_nodes, far_leaves, near_leaves;
        for (size_t l = 0; l < take; ++l)
        {
            const Item it = batch[take - 1 - l];
            m.node_visits += 1;
            Ray ray = r[it.ray];
            ray.t_max = bound[it.ray];
            const float4 *q = sc.walk_nodes + 4 * static_cast<size_t>(it.ref);
            float e0, e1;
            const bool h0 = box_enter(q[0], q[1], ray, e0), h1 = box_enter(q[2], q[3], ray, e1);
            const uint32_t r0 = as_uint(q[0].w), r1 = as_uint(q[1].w);
            const bool first0 = e0 <= e1, both = h0 && h1;
            if (!(h0 || h1))
                continue;
            const uint32_t toward = (h0 && (first0 || !h1)) ? r0 : r1, other = first0 ? r1 : r0;
            if (wide)
            {
                // a step tests two levels: a hit inner child is opened at once and ITS hit children are what is pushed (the
                // work of a 4-wide node: up to four boxes beyond the first two, up to four pushes)
                const uint32_t kids[2] = {unordered ? r0 : other, unordered ? r1 : toward};
                const bool hit[2] = {unordered ? h0 : both, unordered ? h1 : true};
                for (int c = 0; c < 2; ++c)
                {
                    if (!hit[c])
                        continue;
                    if (kids[c] & kWalkLeaf)
                    {
                        (c ? near_leaves : far_leaves).push_back(Item{it.ray, kids[c]});
                        continue;
                    }
                    m.node_visits += 1;
                    const float4 *g = sc.walk_nodes + 4 * static_cast<size_t>(kids[c]);
                    float f0, f1;
                    const bool g0 = box_enter(g[0], g[1], ray, f0), g1 = box_enter(g[2], g[3], ray, f1);
                    const uint32_t s0 = as_uint(g[0].w), s1 = as_uint(g[1].w);
                    const bool gf0 = f0 <= f1;
                    const uint32_t order[2] = {gf0 ? s1 : s0, gf0 ? s0 : s1};
                    const bool oh[2] = {gf0 ? g1 : g0, gf0 ? g0 : g1};
                    if (unordered)
                    {
                        // children in the order the node stores them, whatever the ray's direction
                        const uint32_t ks[2] = {kids[c] == r0 ? s0 : s0, s1};
                        const bool kh[2] = {g0, g1};
                        for (int q = 0; q < 2; ++q)
                            if (kh[q])
                                ((ks[q] & kWalkLeaf) ? near_leaves : near_nodes).push_back(Item{it.ray, ks[q]});
                        continue;
                    }
                    for (int q = 0; q < 2; ++q)
                        if (oh[q])
                            ((order[q] & kWalkLeaf) ? (c ? near_leaves : far_leaves) : (c ? near_nodes : far_nodes)).push_back(Item{it.ray, order[q]});
                }
                continue;
            }
            ((toward & kWalkLeaf) ? near_leaves : near_nodes).push_back(Item{it.ray, toward});
            if (both)
                ((other & kWalkLeaf) ? far_leaves : far_nodes).push_back(Item{it.ray, other});
        }
        // far children below near children: the next step takes the near ones first
        nodes.insert(nodes.end(), far_nodes.begin(), far_nodes.end()), nodes.insert(nodes.end(), near_nodes.begin(), near_nodes.end());
        leaves.insert(leaves.end(), far_leaves.begin(), far_leaves.end()), leaves.insert(leaves.end(), near_leaves.begin(), near_leaves.end());
    }
    for (size_t i = 0; i < nc; ++i)
        m.accepted += accepted[i], m.rays += 1, m.accepted_hist[std::min<uint32_t>(accepted[i], 7)] += 1;
}

// params: {merged (0 / 1), prim_at, wide (0 / 1: two levels per step)};  out: Pool2Model's fields in order (23 doubles) preceded by the per-lane model's
// {wave node steps, lane node visits, wave prim phases, lane prim tests}.
int mcpt_emu_pool_model2(const char *mcsd_path, const uint32_t *params, double *out)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        if (flat.integrator.has_masks || (flat.features & kFeatAnalytic))
            throw std::runtime_error("triangle-only scenes without masks");
        using C = Config<kFeatVolPath | kFeatEmitters | kFeatTextures | kFeatMicrofacet | kFeatOrderedWalk>;
        const uint32_t w = sc.camera.width, h = sc.camera.height;
        const uint32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
        const bool merged = params[0] != 0, wide = params[2] != 0, unordered = params[3] != 0;
        const uint32_t prim_at = params[1];
        std::vector<double> acc(27, 0.0);
        std::mutex mu;
        std::atomic<uint32_t> next{0};
        auto work = [&]()
        {
            double a[4] = {0, 0, 0, 0};
            Pool2Model pm;
            for (;;)
            {
                const uint32_t tile = next.fetch_add(1);
                if (tile >= tx * ty)
                    break;
                PathState st[64];
                std::vector<LaneCounters> cnt(64);
                std::vector<uint32_t> stacks(64 * kWalkStackMax);
                bool has[64];
                for (uint32_t l = 0; l < 64; ++l)
                {
                    const uint32_t x = (tile % tx) * 8 + (l & 7), y = (tile / tx) * 8 + (l >> 3);
                    has[l] = x < w && y < h;
                    cnt[l] = LaneCounters{};
                    st[l].stack = &stacks[l * kWalkStackMax];
                    if (has[l])
                        start_pixel(st[l], y * w + x);
                }
                for (;;)
                {
                    bool any = false;
                    std::vector<std::array<float, 7>> closest, shadow;
                    uint32_t mcn = 0, mcp = 0, msn = 0, msp = 0;
                    for (uint32_t l = 0; l < 64; ++l)
                    {
                        if (!has[l])
                            continue;
                        if (!st[l].alive)
                        {
                            if (st[l].sample >= sc.camera.spp)
                            {
                                has[l] = false;
                                continue;
                            }
                            start_sample(sc, st[l]);
                        }
                        any = true;
                        path_step<C>(sc, st[l], &cnt[l]);
                        std::array<float, 7> rec;
                        std::copy(cnt[l].last_closest_ray, cnt[l].last_closest_ray + 7, rec.begin());
                        closest.push_back(rec);
                        if (cnt[l].last_shadow_count)
                        {
                            std::copy(cnt[l].last_shadow_ray, cnt[l].last_shadow_ray + 7, rec.begin());
                            shadow.push_back(rec);
                        }
                        const uint32_t cn = cnt[l].last_closest_nodes / 2, cp = cnt[l].last_closest_prims;
                        const uint32_t sn = cnt[l].last_shadow_nodes / 2, sp = cnt[l].last_shadow_prims;
                        a[1] += cn + sn, a[3] += cp + sp;
                        mcn = std::max(mcn, cn), mcp = std::max(mcp, cp), msn = std::max(msn, sn), msp = std::max(msp, sp);
                    }
                    if (!any)
                        break;
                    a[0] += mcn + msn, a[2] += mcp + msp;
                    if (merged)
                        RunPool2(sc, closest, shadow, prim_at, pm, wide, unordered);
                    else
                    {
                        RunPool2(sc, closest, {}, prim_at, pm, wide, unordered);
                        RunPool2(sc, {}, shadow, prim_at, pm, wide, unordered);
                    }
                }
            }
            std::lock_guard<std::mutex> lock(mu);
            for (int i = 0; i < 4; ++i)
                acc[i] += a[i];
            const double f[23] = {pm.node_steps, pm.node_visits, pm.prim_steps, pm.prim_tests, pm.rounds, 0, 0, pm.accepted, pm.rays,
                                  pm.accepted_hist[0], pm.accepted_hist[1], pm.accepted_hist[2], pm.accepted_hist[3], pm.accepted_hist[4],
                                  pm.accepted_hist[5], pm.accepted_hist[6], pm.accepted_hist[7], pm.node_lanes_hist[0], pm.node_lanes_hist[1],
                                  pm.node_lanes_hist[2], pm.node_lanes_hist[3], 0, 0};
            for (int i = 0; i < 23; ++i)
                acc[4 + i] += f[i];
            acc[4 + 5] = std::max(acc[4 + 5], pm.max_nodes), acc[4 + 6] = std::max(acc[4 + 6], pm.max_leaves);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < std::max(1u, std::thread::hardware_concurrency()); ++t)
            pool.emplace_back(work);
        work();
        for (std::thread &t : pool)
            t.join();
        for (int i = 0; i < 27; ++i)
            out[i] = acc[i];
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// out: per-lane model {node steps (wave), node visits (lane), prim phases, prim tests} followed by the
// pooled model's {node steps, node visits, prim steps, prim tests}; triangle-only scenes.
int mcpt_emu_pool_model(const char *mcsd_path, double *out)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        if (flat.integrator.has_masks || (flat.features & kFeatAnalytic))
            throw std::runtime_error("triangle-only scenes without masks");
        using C = Config<kFeatVolPath | kFeatEmitters | kFeatTextures | kFeatMicrofacet | kFeatOrderedWalk>;
        const uint32_t w = sc.camera.width, h = sc.camera.height;
        const uint32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
        std::vector<double> acc(24, 0.0);
        std::mutex mu;
        std::atomic<uint32_t> next{0};
        auto work = [&]()
        {
            std::vector<double> a(8 + 16, 0.0);
            PoolModel pm, sm[4];
            for (;;)
            {
                const uint32_t tile = next.fetch_add(1);
                if (tile >= tx * ty)
                    break;
                PathState st[64];
                std::vector<LaneCounters> cnt(64);
                std::vector<uint32_t> stacks(64 * kWalkStackMax);
                bool has[64];
                for (uint32_t l = 0; l < 64; ++l)
                {
                    const uint32_t x = (tile % tx) * 8 + (l & 7), y = (tile / tx) * 8 + (l >> 3);
                    has[l] = x < w && y < h;
                    cnt[l] = LaneCounters{};
                    st[l].stack = &stacks[l * kWalkStackMax];
                    if (has[l])
                        start_pixel(st[l], y * w + x);
                }
                for (;;)
                {
                    bool any = false;
                    std::vector<std::array<float, 7>> closest, shadow;
                    uint32_t mcn = 0, mcp = 0, msn = 0, msp = 0;
                    for (uint32_t l = 0; l < 64; ++l)
                    {
                        if (!has[l])
                            continue;
                        if (!st[l].alive)
                        {
                            if (st[l].sample >= sc.camera.spp)
                            {
                                has[l] = false;
                                continue;
                            }
                            start_sample(sc, st[l]);
                        }
                        any = true;
                        path_step<C>(sc, st[l], &cnt[l]);
                        std::array<float, 7> rec;
                        std::copy(cnt[l].last_closest_ray, cnt[l].last_closest_ray + 7, rec.begin());
                        closest.push_back(rec);
                        if (cnt[l].last_shadow_count)
                        {
                            std::copy(cnt[l].last_shadow_ray, cnt[l].last_shadow_ray + 7, rec.begin());
                            shadow.push_back(rec);
                        }
                        const uint32_t cn = cnt[l].last_closest_nodes / 2, cp = cnt[l].last_closest_prims;
                        const uint32_t sn = cnt[l].last_shadow_nodes / 2, sp = cnt[l].last_shadow_prims;
                        a[1] += cn + sn, a[3] += cp + sp;
                        mcn = std::max(mcn, cn), mcp = std::max(mcp, cp), msn = std::max(msn, sn), msp = std::max(msp, sp);
                    }
                    if (!any)
                        break;
                    a[0] += mcn + msn, a[2] += mcp + msp;
                    RunPool(sc, closest, false, pm);
                    RunPool(sc, shadow, true, pm);
                    for (uint32_t spec = 0; spec < 4; ++spec)
                    {
                        RunSpeculative(sc, closest, false, spec, sm[spec]);
                        RunSpeculative(sc, shadow, true, spec, sm[spec]);
                    }
                }
            }
            a[4] = pm.node_steps, a[5] = pm.node_visits, a[6] = pm.prim_steps, a[7] = pm.prim_tests;
            for (int k = 0; k < 4; ++k)
                a[8 + 4 * k] = sm[k].node_steps, a[9 + 4 * k] = sm[k].node_visits, a[10 + 4 * k] = sm[k].prim_steps,
                          a[11 + 4 * k] = sm[k].prim_tests;
            std::lock_guard<std::mutex> lock(mu);
            for (int i = 0; i < 24; ++i)
                acc[i] += a[i];
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < std::max(1u, std::thread::hardware_concurrency()); ++t)
            pool.emplace_back(work);
        work();
        for (std::thread &t : pool)
            t.join();
        for (int i = 0; i < 24; ++i)
            out[i] = acc[i];
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// Debug: the steps of one pixel with either walk: the CPU build of hip/render_kernel.hip's
// trace_pixel_kernel, same 16-float record per step (mcpt.h, mcpt_debug_trace_pixel).
int mcpt_emu_debug_pixel(const char *mcsd_path, uint32_t pixel, int ordered, float *out, uint32_t capacity)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        constexpr uint32_t kAllF = kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet;
        PathState st;
        uint32_t stack[kWalkStackMax];
        st.stack = stack;
        LaneCounters cnt{};
        start_pixel(st, pixel);
        uint32_t n = 0;
        while (!pixel_done(sc, st) && n < capacity)
        {
            if (!st.alive)
                start_sample(sc, st);
            if (ordered)
                path_step<Config<kAllF | kFeatOrderedWalk | kFeatSlivers>>(sc, st, &cnt);
            else
                path_step<Config<kAllF>>(sc, st, &cnt);
            float *o = out + 16 * n++;
            for (int k = 0; k < 6; ++k)
                o[k] = cnt.last_closest_ray[k];
            o[6] = max_component(st.throughput);
            o[7] = static_cast<float>(cnt.last_hit_prim == kNone ? -1.0 : double(cnt.last_hit_prim));
            o[8] = cnt.last_hit_t, o[9] = static_cast<float>(cnt.last_shadow_count);
            o[10] = static_cast<float>(cnt.last_shadow_hit);
            std::memcpy(&o[11], &st.rng, 4);
            o[12] = st.L.x, o[13] = st.L.y, o[14] = st.L.z, o[15] = static_cast<float>(st.depth);
        }
        return static_cast<int>(n);
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return -1;
    }
}

// Closest-hit queries with either walk, one per ray (6 floats each): out = 2 floats per ray
// {global primitive or -1, distance}.  Multi-threaded over the rays.
int mcpt_emu_closest(const char *mcsd_path, const float *rays, uint32_t n, int ordered, float *out)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        const unsigned n_threads = std::max(1u, std::thread::hardware_concurrency());
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t)
            pool.emplace_back(
                [&, t]()
                {
                    std::vector<uint32_t> stack(kWalkStackMax * kWalkStackStride);
                    for (uint32_t i = t; i < n; i += n_threads)
                    {
                        Ray ray = make_ray(V3{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]},
                                           V3{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
                        HitRaw raw;
                        TraceStats ts{0, 0, 0, 0};
                        uint32_t rng = 1;
                        const bool hit = ordered == 2   ? walk_wide_vote<false, true, false>(sc, stack.data(), ray, raw, ts) // the 4-wide quantised hierarchy
                                         : ordered ? walk_ordered<false, true, false>(sc, stack.data(), ray, raw, ts)
                                                   : walk_scene<false, true, true, false>(sc, ray, rng, raw, ts);
                        out[2 * i] = hit ? static_cast<float>(raw.prim) : -1.0f;
                        out[2 * i + 1] = ray.t_max;
                    }
                });
        for (std::thread &th : pool)
            th.join();
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return -1;
    }
}

// Debug: every triangle slot the ray (origin, direction) hits, brute force: 6 floats per
// hit {primitive, instance, rank, distance, leaf box passes with `t_bound`, reference-walk
// distance of a query restricted to nothing — 0}.
int mcpt_emu_probe_ray(const char *mcsd_path, const float *ray7, float t_bound, float *out, uint32_t capacity)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        const Ray ray = make_ray(V3{ray7[0], ray7[1], ray7[2]}, V3{ray7[3], ray7[4], ray7[5]});
        uint32_t n = 0;
        for (uint32_t slot = 0; slot < flat.integrator.n_prims && n < capacity; ++slot)
        {
            const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(slot);
            const uint32_t prim = as_uint(p[0].w), inst = as_uint(p[1].w), rank = as_uint(p[2].w);
            if (sc.instances[inst].kind != kInstTriangles)
                continue;
            const SlotHit h = triangle_probe(p, ray);
            if (!h.hit)
                continue;
            float *o = out + 6 * n++;
            o[0] = float(prim), o[1] = float(inst), o[2] = float(rank), o[3] = h.t;
            o[4] = reference_leaf_box_passes<true>(sc, inst, prim, ray, t_bound) ? 1.0f : 0.0f;
            o[5] = 0;
        }
        return static_cast<int>(n);
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return -1;
    }
}

// The ordered-walk hierarchy of the product's host commit: 16 floats per node,
// 12 per primitive slot (bit patterns preserved), and (n_nodes, n_slots, depth,
// has_masks) in `counts`.  Returns 0, or the required capacities in counts on -2.
int mcpt_emu_walk(const char *mcsd_path, float *nodes, float *prims, uint32_t node_capacity, uint32_t slot_capacity,
                  uint32_t *counts)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const uint32_t n_nodes = flat.integrator.n_walk_nodes;
        const uint32_t n_slots = n_nodes ? flat.integrator.n_prims : 0;
        counts[0] = n_nodes, counts[1] = n_slots, counts[2] = flat.integrator.walk_depth;
        counts[3] = flat.integrator.has_masks;
        counts[4] = static_cast<uint32_t>(flat.seconds_lbvh * 1e3), counts[5] = static_cast<uint32_t>(flat.seconds_walk * 1e3);
        counts[6] = static_cast<uint32_t>(flat.seconds_total * 1e3);
        if (n_nodes > node_capacity || n_slots > slot_capacity)
            return -2;
        std::memcpy(nodes, flat.walk_nodes.data(), size_t(n_nodes) * 64);
        std::memcpy(prims, flat.walk_prims.data(), size_t(n_slots) * 48);
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return -1;
    }
}

// The 4-wide exact form of the ordered-walk hierarchy (DeviceScene::pool_nodes): 32 words per node.  counts: nodes, depth.
int mcpt_emu_pool_nodes(const char *mcsd_path, float *nodes, uint32_t node_capacity, uint32_t *counts)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        counts[0] = flat.integrator.n_pool_nodes, counts[1] = flat.integrator.pool_depth;
        if (flat.integrator.n_pool_nodes > node_capacity)
            return -2;
        std::memcpy(nodes, flat.pool_nodes.data(), size_t(flat.integrator.n_pool_nodes) * 128);
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return -1;
    }
}

// Committed tables of the product's host commit, for comparison with the
// oracle's: nodes as (skip, object) u32 pairs + 6 box floats + area.
int mcpt_emu_nodes(const char *mcsd_path, uint32_t *links, float *geom, uint32_t capacity)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const uint32_t n = static_cast<uint32_t>(flat.node_area.size());
        if (n > capacity)
            return static_cast<int>(n);
        for (uint32_t i = 0; i < n; ++i)
        {
            const float4 a = flat.nodes[2 * i], b = flat.nodes[2 * i + 1];
            std::memcpy(&links[2 * i], &a.w, 4), std::memcpy(&links[2 * i + 1], &b.w, 4);
            geom[7 * i] = flat.node_area[i];
            geom[7 * i + 1] = a.x, geom[7 * i + 2] = a.y, geom[7 * i + 3] = a.z;
            geom[7 * i + 4] = b.x, geom[7 * i + 5] = b.y, geom[7 * i + 6] = b.z;
        }
        return static_cast<int>(n);
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return -1;
    }
}

} // extern "C"
