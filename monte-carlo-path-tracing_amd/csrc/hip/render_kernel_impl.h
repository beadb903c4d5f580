// HIP render kernel for gfx950 (MI355X): persistent lanes running the streaming
// path-tracing state machine of path_core.h.
//
// Launch shape.  A workgroup is 256 lanes = 4 wavefronts of 64.  Work items are
// pixels enumerated tile by tile (8x8 pixels per tile, so one wavefront starts
// on one coherent tile); a lane processes item q, q + stride, ... where stride
// is the total number of launched lanes, so the grid is sized to the machine
// (CUs x resident workgroups) rather than to the image, and every lane keeps
// regenerating paths until its pixels are exhausted.  No data is exchanged
// between lanes: the per-pixel sequential RNG chain of the reference makes
// the pixel the unit of parallelism.
//
// Memory.  All scene tables are read-only arrays in HBM (device_scene.h).  A ray
// query reads one 64-byte two-box node (four float4 loads) per step and one 48-byte
// primitive slot per leaf; hit attributes (144 B) are read once per shaded hit.
// Path state lives in registers, the traversal stacks in LDS (lane-interleaved);
// the only global writes are 12 B per finished pixel.  Scenes whose traversal data
// fits 24 KiB are staged into LDS per workgroup, larger ones are read through
// L1 / L2.  There is no matrix-shaped work here: no MFMA.
//
// This file is the shared part of the kernel translation units: the kernel template and its
// launcher.  The instantiations are spread over render_kernel.hip (lean variants, unit kernels,
// the dispatcher), render_variants_all.hip, render_variants_counted.hip and
// render_variants_surface.hip so that `make -j` compiles them side by side (one unit took 3.7
// minutes); the `extern template` lines below keep a unit from instantiating another unit's kernels.
#ifndef MCPT_RENDER_KERNEL_IMPL_H
#define MCPT_RENDER_KERNEL_IMPL_H

#include "../wave_target.h"
#if !defined(MCPT_WAVE_EMU)
#include <hip/hip_runtime.h>
#endif

#include "../path_core.h"
#include "render_kernel.h"

namespace mcpt
{

// Register budget: the lean instantiations (no microfacet / volume code) are
// held to 128 VGPRs = 4 wavefronts per SIMD = 4 workgroups of 256 per CU, so
// that a 512x512 frame (262 144 pixels = 256 CUs x 1024 lanes) is resident in
// one round; the material instantiations run 3 per SIMD.
template <uint32_t kFeatures, bool kLdsGeometry = true>
struct Budget
{
    // The lean instantiations for scenes too large for LDS are memory-latency bound (0.8 M
    // triangles: 63 % of wave cycles waiting, VALU pipe 41 % busy): held to 6 per SIMD (80 VGPRs,
    // some spilling) they are 11 % (blob field) and 19 % (terrain) faster than at 4.
    // measured: the full instantiation at 3 per SIMD (<= 168 VGPRs) is 17 % faster on the
    // volumetric scenes than at 2 (208 VGPRs), slower again at 4 (-3 %) and 6 (-24 %); the
    // surface-materials instantiation (matpreview) is fastest at 6 (rough dielectric +25 %,
    // rough conductor +3 % over its natural 168 VGPRs)
    // (the same instantiation with the hierarchy in LDS, volumetric-caustic at 2 / 3 / 4 per SIMD: 779 / 913 / 890 Msamples/s)
#ifndef MCPT_FULL_LDS_WAVES
#define MCPT_FULL_LDS_WAVES 3
#endif
    // (pool walk outside LDS: a wavefront's pool area is 9.5 KB of LDS, 16 of them fit a CU.  Measured at 4 / 3 wavefronts per
    //  SIMD — 128 VGPRs with 49-140 spilled / 156-168 with none: dragon/scene.xml 239.5 / 230.8 ms, matpreview rough conductor
    //  652.0 / 613.6 ms, rough dielectric 902.8 / 905.9 ms)
    // (the one-BSDF units without merged queries — matpreview — need 129-131 VGPRs left alone and are throughput-bound: held to 128
    //  = 4 wavefronts per SIMD they spill nothing, rough dielectric 849 -> 739 ms; merged queries' larger pool areas would cost them
    //  that fourth wavefront, csrc/Makefile.  -DMCPT_POOL_BIG_WAVES=<n> overrides all of them in experiment builds)
#ifdef MCPT_POOL_BIG_WAVES
    static constexpr int kPoolBigWaves = MCPT_POOL_BIG_WAVES;
#else
    static constexpr int kPoolBigWaves = (kFeatures & (kFeatConductorOnly | kFeatDielectricOnly)) && !(kFeatures & kFeatPoolMerge) ? 4 : 3;
#endif
    static constexpr int kWavesPerSimd = (kFeatures & kFeatPoolBig) ? kPoolBigWaves
                                         : (kFeatures & (kFeatVolPath | kFeatAnalytic)) ? (kLdsGeometry ? MCPT_FULL_LDS_WAVES : 3)
#ifdef MCPT_EXPERIMENT_MICROFACET_WAVES
                                         : (kFeatures & kFeatMicrofacet)              ? MCPT_EXPERIMENT_MICROFACET_WAVES
#else
                                         : (kFeatures & kFeatMicrofacet)              ? 6
#endif
                                         : kLdsGeometry                                ? 4
                                                                                      : 6;
};

#ifndef MCPT_POOL_MAX_SPREAD
#define MCPT_POOL_MAX_SPREAD 16
#endif
constexpr uint32_t kPoolMaxSpread = MCPT_POOL_MAX_SPREAD; // lanes per path at most, when a launch has fewer pixels than lanes (pool walk)
constexpr uint32_t kFetchNext = 0xFFFFFFFFu; // a lane's item variable: "ask the work counter when the current pixel is done"
constexpr uint32_t kCompactWords = 8, kCompactPasses = 4; // a path's 30 state words travel through LDS in four passes of eight
constexpr uint32_t kSpreadPasses = 6; // ... with a pending shadow ray (merged queries: 17 words more) in six
#ifndef MCPT_TAIL_SPREAD
#define MCPT_TAIL_SPREAD 1
#endif

// kLdsGeometry: the arrays the ray queries and the light sampler read (both
// hierarchies, walk primitives, triangle positions) are copied into LDS by each
// workgroup before it starts and read from there (ds_read_b128) instead of
// through L1.  Used for scenes whose traversal data fits kLdsGeometryBytes
// (cornell: 8 KB); a walk is a chain of dependent loads, so the shorter LDS
// latency shortens every step.  Large scenes stream from HBM / L2.
// The ordered walk's stacks always live in LDS: lane t of the workgroup owns the
// words t, t + 256, t + 512, ... of the stack area.
// The XCD this wavefront runs on (0-7; observed: workgroup b runs on XCD b % 8, MI355X_MICROARCH guide — a performance hint only).
__device__ __forceinline__ uint32_t xcd_id()
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return static_cast<uint32_t>(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20)) & (kBands - 1u);
}

// RenderJob::xcd_bands: the next hand-out position for every calling lane — from band `first` while it has items, then from the bands
// behind it (a band that ran dry is skipped with one load).  n_work: when nothing is left anywhere.  Bands are whole tiles.
__device__ __forceinline__ uint32_t band_reserve(uint32_t *counters, uint32_t first, uint32_t n_work)
{
    const uint32_t tiles = n_work >> 6;
    uint32_t got = n_work;
    bool want = true;
    for (uint32_t k = 0; k < kBands; ++k) // (uniform: the calling lanes of a wavefront go through the bands together)
    {
        if (__ballot(want) == 0)
            break;
        const uint32_t b = (first + k) & (kBands - 1u);
        const uint32_t begin = (static_cast<uint64_t>(tiles) * b / kBands) << 6, end = (static_cast<uint64_t>(tiles) * (b + 1u) / kBands) << 6;
        uint32_t *counter = counters + kBandStride * b;
        const uint32_t taken = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
        if (taken >= end - begin)
            continue; // (dry: a stale value says "not yet" at worst, and the reservation below finds out)
        const uint32_t r = wave_reserve(counter, want);
        if (want && r < end - begin)
            got = begin + r, want = false;
    }
    return got;
}

template <uint32_t kFeatures, bool kCount, bool kLdsGeometry>
__device__ __forceinline__ void render_body(const DeviceScene &sc_in, const RenderJob &job, float *__restrict__ out, TraceCounters *__restrict__ counters)
{
    using C = Config<kFeatures>;
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(((kFeatures & kFeatLowDisc) != 0) == (MCPT_LOW_DISCREPANCY_ACTIVE != 0), "low-discrepancy instantiations live in their own translation unit");
#endif
    MCPT_DYNAMIC_LDS(float4, lds_geometry);
    if (job.wave_clock && (threadIdx.x & 63u) == 0)
        job.wave_clock[4u * (blockIdx.x * (kBlockSize / 64u) + (threadIdx.x >> 6))] = wall_clock64();
#if MCPT_PHASE_CLOCK
    if ((threadIdx.x & 63u) == 0) // (diagnostic builds: path_core.h, phase_mark)
    {
        unsigned long long *a = phase_area();
        for (uint32_t k = 1; k < 1 + 3 * kPhaseCount; ++k)
            a[k] = 0;
        a[0] = clock64();
    }
#endif
    DeviceScene sc = sc_in;
    uint32_t n_staged = 0;
    if (kLdsGeometry)
    {
        const uint32_t n_node_vec = 2u * sc_in.integrator.n_nodes, n_tri_vec = 3u * sc_in.integrator.n_prims;
        // (pool walk: the 4-wide exact form of the hierarchy instead of the binary one)
        const uint32_t n_walk_vec = C::kPool ? kPoolLdsNodeVecs * sc_in.integrator.n_pool_nodes : C::kOrdered ? 4u * sc_in.integrator.n_walk_nodes : 0u;
        const uint32_t n_slot_vec = C::kOrdered ? n_tri_vec : 0u;
        for (uint32_t i = threadIdx.x; i < n_node_vec; i += blockDim.x)
            lds_geometry[i] = sc_in.nodes[i];
        for (uint32_t i = threadIdx.x; i < n_tri_vec; i += blockDim.x)
            lds_geometry[n_node_vec + i] = sc_in.tri_pos[i];
        // (pool walk: the staged node records lie kPoolLdsNodeVecs vectors apart — LDS banks, pool_walk.h)
        for (uint32_t i = threadIdx.x; i < (C::kPool ? 8u * sc_in.integrator.n_pool_nodes : n_walk_vec); i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + (C::kPool ? (i >> 3) * kPoolLdsNodeVecs + (i & 7u) : i)] = C::kPool ? sc_in.pool_nodes[i] : sc_in.walk_nodes[i];
        for (uint32_t i = threadIdx.x; i < n_slot_vec; i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + n_walk_vec + i] = sc_in.walk_prims[i];
        __syncthreads();
        sc.nodes = lds_geometry;
        sc.tri_pos = lds_geometry + n_node_vec;
        if (C::kOrdered)
        {
            // (whichever form was staged; pool walk: the binary form stays where it is, in HBM, for the rare ray that is walked
            //  the per-lane way — pool_walk.h)
            if (C::kPool)
                sc.pool_nodes = lds_geometry + n_node_vec + n_tri_vec;
            else
                sc.walk_nodes = lds_geometry + n_node_vec + n_tri_vec;
            sc.walk_prims = lds_geometry + n_node_vec + n_tri_vec + n_walk_vec;
        }
        n_staged = n_node_vec + n_tri_vec + n_walk_vec + n_slot_vec;
    }
    // (lane_spread is a power of two up to 64: the lanes 0, spread, 2 spread, ... of a wavefront take items)
    const uint32_t spread = job.lane_spread ? job.lane_spread : 1u;
    const uint32_t stride = gridDim.x * blockDim.x / spread;
    const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t q = lane_id / spread;
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);

    LaneCounters local{};
    LaneCounters *cnt = kCount ? &local : nullptr;

    PathState st{}; // (every field defined: a lane that has not started a pixel yet can be moved by a compaction)
    st.alive = false;
    PendingShadow pend{}; // merged queries (path_step_merged): the last vertex's shadow ray, waiting for the next walk
    pend.shadow = pend.finish = false;
    // (pool walk: the wavefront's pool area instead of the lane's stack column)
    // (the wavefront's number as a scalar: every address inside its pool area is then a scalar base + a lane offset)
    st.stack = C::kPool ? reinterpret_cast<uint32_t *>(lds_geometry + n_staged) +
                              static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6))) * pool_wave_words(C::kAnalytic, C::kPoolBig, C::kPoolDual)
                        : reinterpret_cast<uint32_t *>(lds_geometry + n_staged) + threadIdx.x;
    bool has_pixel = false;
    uint32_t slot = 0; // where this pixel's result goes
    uint32_t steps = 0, my_tile = 0; // cost probe (RenderJob::tile_steps)
    // EVENTS OF A THINNING WORKGROUP ("COMPACTION": LDS-resident scenes).  Once the job's items are handed out, a lane whose pixel is
    // finished has nothing left to do.  Whenever another 64 lanes of the workgroup have retired, its four wavefronts meet at a barrier
    // and the paths still in flight move — through LDS, state word by state word.  Round 3 PACKED them into the first wavefronts of the
    // workgroup (the kernel is VALU-issue bound, an instruction of a wavefront with 20 live lanes costs what one with 64 does: cornell
    // 1074 -> 1098 Msamples/s then); since the pool walk's lanes without a path help their wavefront's queries and node steps with few
    // items take 4 / 2 lanes per item, the opposite wins and the events DEAL the paths out (below, MCPT_COMPACT_DEALS: cornell 37.3 ->
    // 36.9 ms).  A path's state is all that makes its pixel (RNG, sample counter, ray, sums): which lane carries it is irrelevant,
    // the frame is unchanged.
    // (the full-feature instantiation — volumetric-caustic, 3.5 pixels per lane from the work counter — 1012 -> 1007 with packing: not there)
    constexpr bool kCompact = kLdsGeometry && !kCount && C::kOrdered && !(C::kVolPath || C::kAnalytic) && !C::kPoolDual; // (the compaction carries no pending shadow ray)
    // TAIL SPREAD (round 6; pool-walk kernels outside LDS with the work counter: RenderJob::tail_spread).  The same events, the other
    // way round.  dragon/scene.xml's frame ends on a few dozen wavefronts that hold 32 paths of its most expensive tiles each — a tile
    // alone takes 95 ms at 1 path per 2 lanes, 52 at 1 per 8 (EXPERIMENTS R6-10) — while the other wavefronts of their workgroups
    // have nothing left and wait.  Once a wavefront of the workgroup has been refused by the work counter (dry: nothing will ever be
    // handed out again), lanes without a pixel retire, and whenever the workgroup is down to 96 / 64 / 32 / 16 / 8 paths these are
    // DEALT OUT over its four wavefronts — path j of the workgroup's order to wavefront j mod 4 —, pending shadow rays included: every
    // wavefront then holds a quarter of the paths and three quarters more helper lanes.  A path's state is all that makes its pixel:
    // the frame is unchanged.
    // Built into the diffuse (+ emitters, slivers) kernels outside LDS — dragon/scene.xml's class: 119.4 -> 107.1 ms (108.9-131.6 ->
    // 101.5-115.1 over 24 draws), its 1/8 share 44.5 -> 40.1.  The one-BSDF surface units lose 1.5-1.9 % with the code compiled in
    // (they sit at their 128-register limit, and their tails are what lanes per path by tile cost is for): not there (EXPERIMENTS R6-13).
    // (-DMCPT_TAIL_SPREAD=2, experiment builds: in every surface instantiation outside LDS — with render_kernel.hip's TailSpreadRuns)
    constexpr bool kTailSpread = MCPT_TAIL_SPREAD != 0 && C::kPool && C::kPoolBig && !kLdsGeometry && !kCount &&
                                 (kFeatures & (kFeatVolPath | kFeatAnalytic | (MCPT_TAIL_SPREAD == 2 ? 0u : kFeatMicrofacet | kFeatTextures))) == 0;
    // (-DMCPT_LDS_MARKET=1, experiment builds: the LDS kernels with the compaction take part in the path market too)
#ifndef MCPT_LDS_MARKET
#define MCPT_LDS_MARKET 0
#endif
    constexpr bool kMarket = kTailSpread || (kCompact && C::kPool && MCPT_LDS_MARKET != 0);
    constexpr bool kEventsBuilt = kCompact || kTailSpread;
    constexpr uint32_t kEvents = kTailSpread ? 5u : kBlockSize / 64u - 1u;
    // (event k falls due when this many lanes of the workgroup have retired)
    auto event_at = [](uint32_t k) { return kTailSpread ? (k == 0 ? 160u : k == 1 ? 192u : k == 2 ? 224u : k == 3 ? 240u : 248u) : 64u * (k + 1u); };
    constexpr uint32_t kEventPasses = kTailSpread && C::kPoolDual ? kSpreadPasses : kCompactPasses;
    const bool events_on = kEventsBuilt && (kTailSpread ? job.tail_spread != 0 && job.work_counter != nullptr && job.tile_steps == nullptr && job.xcd_bands == 0 : job.compact != 0);
    // (pool walk: the words travel through the pool areas — no wavefront is inside a query during an event — and the
    //  counters, which are read at every step, have their own words behind them)
    uint32_t *compact_words = C::kPool ? reinterpret_cast<uint32_t *>(lds_geometry + n_staged)
                                       : reinterpret_cast<uint32_t *>(lds_geometry + n_staged) + static_cast<size_t>(sc_in.integrator.walk_depth) * kBlockSize;
    uint32_t *compact_count = C::kPool ? reinterpret_cast<uint32_t *>(lds_geometry + n_staged) + (kBlockSize / 64u) * pool_wave_words(C::kAnalytic, C::kPoolBig, C::kPoolDual)
                                       : compact_words + kCompactWords * kBlockSize; // [0..3]: live lanes per wavefront, [4]: retired lanes of the workgroup, [5]: the work counter is dry
    bool retired = false;
    uint32_t compact_events = 0; // events this wavefront has taken part in
    if (kEventsBuilt)
    {
        if (threadIdx.x < 8)
            compact_count[threadIdx.x] = 0;
        __syncthreads();
    }
    // reference RNG mode: split == 1, one item per pixel.  Independent-sample mode: item q = k * n_items + pixel
    // item, samples k, k + split, ... (all uniform values: no cost in the reference mode)
    const uint32_t split = job.sample_split ? job.sample_split : 1u, n_work = job.n_items * split;
    const bool independent = job.independent_samples != 0;
    if (lane_id % spread != 0)
        q = n_work;
    // lanes per path by tile cost (RenderJob::level_until): every item comes from the work counter, the first ones too — which lanes
    // may take one depends on what is being handed out
    // (kernels outside LDS only: the LDS kernels' compaction retires a lane that waits for its turn, and moves paths to other lanes)
    const bool levels = C::kPoolBig && job.work_counter != nullptr && job.level_until[2] != 0;
    // XCD bands (RenderJob::xcd_bands): every item comes from the counters too, the band of this workgroup's XCD first
    const bool bands = C::kPoolBig && job.work_counter != nullptr && job.xcd_bands != 0;
    const uint32_t my_band = bands ? xcd_id() : 0u;
    const uint32_t counter_base = levels ? 0u : stride;
    uint32_t my_lg = 0; // log2 of the lanes per path this lane's pixel was handed out with
    if ((levels || bands) && q < n_work)
        q = kFetchNext;
    // a path's state as words (the events' exchange through LDS, the market's records in device memory): 32 words, 48 with the pending
    // shadow ray of merged queries
    constexpr uint32_t kStateWords = kCompactWords * kEventPasses;
    auto pack_path = [&](uint32_t *in) __attribute__((always_inline))
    {
        in[0] = st.rng, in[1] = st.pixel, in[2] = st.sample, in[3] = st.depth;
        in[4] = (st.alive ? 1u : 0u) | (st.primary ? 2u : 0u) | (st.in_medium ? 4u : 0u) | (has_pixel ? 8u : 0u) | (pend.shadow ? 16u : 0u) | (pend.finish ? 32u : 0u) | (my_lg << 8);
        in[5] = st.medium;
        in[6] = as_uint(st.pdf_sample);
        auto put = [&](uint32_t at, V3 v) { in[at] = as_uint(v.x), in[at + 1] = as_uint(v.y), in[at + 2] = as_uint(v.z); };
        put(7, st.origin), put(10, st.dir), put(13, st.wo), put(16, st.wi), put(19, st.throughput), put(22, st.L), put(25, st.pixel_sum);
        in[28] = slot, in[29] = q, in[30] = 0, in[31] = 0;
        if constexpr (kEventPasses > kCompactPasses)
        {
            // (merged queries: the vertex's pending shadow ray and what the sample gains with either answer travel with the path)
            put(30, pend.origin), put(33, pend.dir), in[36] = as_uint(pend.t_max);
            put(37, pend.add_visible), put(40, pend.add_occluded), put(43, pend.old_L);
            in[46] = 0, in[47] = 0;
        }
    };
    auto unpack_path = [&](const uint32_t *got) __attribute__((always_inline))
    {
        st.rng = got[0], st.pixel = got[1], st.sample = got[2], st.depth = got[3];
        st.alive = (got[4] & 1u) != 0, st.primary = (got[4] & 2u) != 0, st.in_medium = (got[4] & 4u) != 0;
        has_pixel = (got[4] & 8u) != 0;
        st.medium = got[5], st.pdf_sample = as_float(got[6]);
        auto get = [&](uint32_t at) { return V3{as_float(got[at]), as_float(got[at + 1]), as_float(got[at + 2])}; };
        st.origin = get(7), st.dir = get(10), st.wo = get(13), st.wi = get(16), st.throughput = get(19), st.L = get(22);
        st.pixel_sum = get(25);
        slot = got[28], q = got[29];
        if constexpr (kEventPasses > kCompactPasses)
        {
            pend.shadow = (got[4] & 16u) != 0, pend.finish = (got[4] & 32u) != 0;
            pend.origin = get(30), pend.dir = get(33), pend.t_max = as_float(got[36]);
            pend.add_visible = get(37), pend.add_occluded = get(40), pend.old_L = get(43);
        }
        my_lg = (got[4] >> 8) & 3u;
    };
    // PATH MARKET (RenderJob::market; kernels with the tail spread): what the tail spread does inside a workgroup, between workgroups.
    uint32_t *const market = kMarket && events_on ? job.market : nullptr;
    if (kMarket && market != nullptr && threadIdx.x == 0)
        atomicAdd(&market[96], 1u); // workgroups of this launch that have started (see where a wavefront decides to wait)
    bool market_mode = false, have_ticket = false, finished_item = false; // (this wavefront's workgroup is done: it waits for paths; its ticket; an item ended on this lane)
    uint32_t ticket = 0, rounds = 0, backoff = 1;
    for (;;)
    {
        MCPT_WAVE_CONVERGE();
        if (kMarket && market != nullptr)
        {
            // (items finished since the last time here: the count a waiting wavefront leaves by)
            const unsigned long long fin = __ballot(finished_item);
            finished_item = false;
            if (fin != 0 && pool_rank(fin) == 0 && (fin >> (threadIdx.x & 63u)) & 1ull)
                atomicAdd(&market[64], static_cast<uint32_t>(__popcll(fin)));
        }
        // WORK COUNTER: a lane that needs a new item takes the first one nobody has taken yet — NOW, when it is free, not ahead
        // of time.  (Rounds 2-4 reserved a lane's next item when it STARTED the current one, to hide the atomic's latency: every
        // lane then held one item hostage while it worked on another — at the end of a frame the items waiting behind the longest
        // pixel chains, with the rest of the GPU drained; with the most expensive tiles handed out first, the first TWO items of
        // every lane were fixed at time 0.  Measured with the wavefront clocks (RenderJob::wave_clock, round 5): dragon/scene.xml's
        // counter ran dry at 46 % of the frame and half of the wavefront slots were empty on average; matpreview rough conductor
        // ran its last 40 % on 2 % of its wavefronts.)
        if (job.work_counter)
        {
            bool want = !has_pixel && q == kFetchNext;
            if (levels && __ballot(want) != 0)
            {
                // the sparsest level among the pixels this wavefront holds and the position the counter stands at (an agent-scope
                // load: the vector cache is not coherent; a stale value is a smaller position, i.e. a sparser level — harmless)
                const uint32_t held = __ballot(has_pixel && my_lg >= 3u) ? 3u : __ballot(has_pixel && my_lg >= 2u) ? 2u : __ballot(has_pixel && my_lg >= 1u) ? 1u : 0u;
                const uint32_t at = __hip_atomic_load(job.work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t next = at < job.level_until[0] ? 3u : at < job.level_until[1] ? 2u : at < job.level_until[2] ? 1u : 0u;
                const uint32_t lg = held > next ? held : next;
                want = want && ((threadIdx.x & 63u) & ((1u << lg) - 1u)) == 0u;
            }
            if (want)
            {
                MCPT_WAVE_REGION();
                q = bands ? band_reserve(job.work_counter, my_band, n_work) : counter_base + wave_reserve(job.work_counter, true);
                // (tail spread: a lane that ASKED and got nothing has seen the counter dry — nothing will be handed out any more)
                if (kTailSpread && events_on && q >= n_work)
                    compact_count[5] = 1u;
            }
        }
        // (pool walk: a lane without work of its own stays in the loop as a HELPER of its wavefront's ray queries)
        bool helper = false;
        if (kMarket && market_mode)
        {
            // ---- this wavefront's workgroup is done: it runs paths other wavefronts give away, one at a time, on its first lane ----
            if (__ballot(has_pixel) == 0)
            {
                if (!have_ticket)
                {
                    uint32_t t = 0;
                    if ((threadIdx.x & 63u) == 0)
                        t = atomicAdd(&market[0], 1u);
                    ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t))), have_ticket = true;
                }
                const uint32_t at = ticket & (kMarketSlots - 1u), generation = ticket / kMarketSlots + 1u;
                // (wavefront-uniform decisions from ONE lane's loads: the lanes' own loads may see different moments)
                auto uniform_load = [&](uint32_t *word, int order) __attribute__((always_inline))
                {
                    uint32_t v = 0;
                    if ((threadIdx.x & 63u) == 0)
                        v = order == __ATOMIC_ACQUIRE ? __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
                };
                if (uniform_load(&market[kMarketReadyAt + at], __ATOMIC_ACQUIRE) == generation)
                {
                    have_ticket = false, backoff = 1;
                    if ((threadIdx.x & 63u) == 0)
                    {
                        uint32_t got[kStateWords];
#pragma unroll
                        for (uint32_t k = 0; k < kStateWords; ++k) // (through L2: the vector cache may hold an older record of this slot)
                            got[k] = __hip_atomic_load(&market[kMarketRecordsAt + kMarketRecord * at + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        unpack_path(got);
                        retired = false;
                    }
                }
                else
                {
                    if (uniform_load(&market[64], __ATOMIC_RELAXED) >= n_work)
                        break; // every item of the job is finished: nothing will be given away any more
                    // (3 000 waiting wavefronts that ask every 3 us keep one L2 channel busy with nothing else — measured: the frame
                    //  THREE times as long; a path that waits 100 us for its wavefront loses nothing against a tail of 20 ms)
                    for (uint32_t i = 0; i < backoff; ++i)
                        __builtin_amdgcn_s_sleep(127);
                    backoff = backoff < 32u ? backoff * 2u : 32u;
                    continue;
                }
            }
            helper = !has_pixel;
        }
        else if (kEventsBuilt && events_on)
        {
            // (one LDS word each, read by every lane of the wavefront in the same instruction: uniform — and said so, the lockstep host
            //  build of this body, tests/emu, runs a wavefront's lanes one after the other between cross-lane operations)
            // tail spread: lanes retire only once the work counter is known to be dry (a lane that waits for its turn — lanes per path
            // by tile cost — may still be given a pixel before that); the compaction's lanes all asked, one pixel per lane
            const bool dry = !kTailSpread || __builtin_amdgcn_readfirstlane(static_cast<int>(*static_cast<volatile uint32_t *>(&compact_count[5]))) != 0;
            if (!has_pixel && !retired && q >= n_work && dry)
            {
                retired = true;
                atomicAdd(&compact_count[4], 1u);
            }
            const uint32_t n_retired = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(*static_cast<volatile uint32_t *>(&compact_count[4]))));
            // (every wavefront takes part in every event, also the ones that fall due together with the end: a
            //  wavefront that left early would leave the others waiting at the event's barriers)
            if (n_retired >= kBlockSize && compact_events == kEvents)
            {
                if (kMarket && market != nullptr)
                {
                    // the workgroup is done: its wavefronts wait for other workgroups' paths — IF every workgroup of the launch has
                    // started.  A wavefront that waits holds its slot until the job's last item is finished; were some workgroups
                    // not resident yet (another kernel on the device: a second renderer, another process), they would wait for the
                    // slots of wavefronts that wait for them.  Then it leaves, like before the market existed.
                    uint32_t started = 0;
                    if ((threadIdx.x & 63u) == 0)
                        started = __hip_atomic_load(&market[96], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(started))) != gridDim.x)
                        break;
                    market_mode = true;
                    continue;
                }
                break; // the workgroup is done
            }
            if (compact_events < kEvents && n_retired >= event_at(compact_events))
            {
                // ---- event: every wavefront of the workgroup comes here once per threshold, in the same order ----
                ++compact_events;
                uint32_t n_live;
                // (live = holds a pixel in flight, or an item it has reserved and not started yet)
                const bool live = !retired;
                const uint32_t rank_in_wave = lane_rank_among(live, n_live);
                __syncthreads();
                if ((threadIdx.x & 63u) == 0)
                    compact_count[threadIdx.x >> 6] = n_live;
                __syncthreads();
                uint32_t before = 0, total = 0;
                for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
                {
                    const uint32_t c = compact_count[w];
                    before += w < (threadIdx.x >> 6) ? c : 0u;
                    total += c;
                }
                // compaction: the paths in the workgroup's order fill its first lanes (the first wavefronts are full again); tail
                // spread: path j goes to wavefront j mod 4, lane j / 4 — which lane of a wavefront carries a path is irrelevant to the
                // pool walk, every lane is a worker
                const uint32_t place = before + rank_in_wave;
                // The LDS kernels' events DEAL the paths out too (round 6).  They used to pack them into the workgroup's first wavefronts
                // (round 3: an instruction of a wavefront with 20 live lanes costs what one with 64 does, +2 % then); since the pool
                // walk's lanes without a path help their wavefront's queries (round 4) and node steps with few items take 4 / 2 lanes
                // per item (round 5), four wavefronts with a quarter of the paths each finish them sooner than one full one: cornell
                // 37.3 (pack) / 37.0 (no events) / 36.9 ms (deal), 16 draws each, same box (EXPERIMENTS R6-15).
                // -DMCPT_COMPACT_DEALS=<k>: only from the k-th event on (99: always pack).
#ifndef MCPT_COMPACT_DEALS
#define MCPT_COMPACT_DEALS 1
#endif
                const bool deal = kTailSpread || compact_events >= MCPT_COMPACT_DEALS;
                const uint32_t dst = deal ? (place & 3u) * 64u + (place >> 2) : place;
                const uint32_t mine = deal ? (threadIdx.x & 63u) * 4u + (threadIdx.x >> 6) : threadIdx.x; // the place whose path this lane receives
                uint32_t in[kStateWords], got[kStateWords];
                pack_path(in);
                static_assert(!(kCompact && C::kPoolDual), "the compaction does not carry a pending shadow ray");
#pragma unroll
                for (uint32_t pass = 0; pass < kEventPasses; ++pass)
                {
                    if (live)
#pragma unroll
                        for (uint32_t k = 0; k < kCompactWords; ++k)
                            compact_words[k * kBlockSize + dst] = in[pass * kCompactWords + k];
                    __syncthreads();
#pragma unroll
                    for (uint32_t k = 0; k < kCompactWords; ++k)
                        got[pass * kCompactWords + k] = compact_words[k * kBlockSize + threadIdx.x];
                    __syncthreads();
                }
                retired = mine >= total;
                if (!retired)
                    unpack_path(got);
                else
                {
                    has_pixel = false;
                    q = n_work;
                    pend.shadow = pend.finish = false;
                }
                continue;
            }
            // ---- path market, the giving side: a wavefront with two or more paths, while tickets wait ----
            if (kMarket && market != nullptr && (rounds++ & 3u) == 0u)
            {
                uint32_t tickets = 0, given = 0;
                if ((threadIdx.x & 63u) == 0) // (one lane's loads, then uniform: the decision below must be the wavefront's)
                {
                    tickets = __hip_atomic_load(&market[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    given = __hip_atomic_load(&market[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                tickets = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(tickets)));
                given = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(given)));
                uint32_t n_paths;
                const bool path = has_pixel && !retired;
                const uint32_t path_rank = lane_rank_among(path, n_paths);
                if (tickets > given && n_paths >= 2u)
                {
                    // (a ticket means a wavefront was refused by the counter and its whole workgroup ran out: the counter is dry)
                    if (!dry)
                        compact_count[5] = 1u;
                    uint32_t base = 0xFFFFFFFFu, n_give = 0;
                    if ((threadIdx.x & 63u) == 0)
                    {
                        // `given` never passes `tickets`: every record has a wavefront waiting for it
                        uint32_t expect = given;
                        for (uint32_t tries = 0; tries < 4u; ++tries)
                        {
                            const uint32_t waiting = __hip_atomic_load(&market[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (waiting <= expect)
                                break;
                            const uint32_t want_give = n_paths / 2u < waiting - expect ? n_paths / 2u : waiting - expect;
                            const uint32_t old = atomicCAS(&market[32], expect, expect + want_give);
                            if (old == expect)
                            {
                                base = expect, n_give = want_give;
                                break;
                            }
                            expect = old;
                        }
                    }
                    base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
                    n_give = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(n_give)));
                    if (n_give != 0 && path && path_rank >= n_paths - n_give)
                    {
                        const uint32_t record = base + (path_rank - (n_paths - n_give)), at = record & (kMarketSlots - 1u);
                        uint32_t in[kStateWords];
                        pack_path(in);
#pragma unroll
                        for (uint32_t k = 0; k < kStateWords; ++k)
                            market[kMarketRecordsAt + kMarketRecord * at + k] = in[k];
                        __threadfence();
                        __hip_atomic_store(&market[kMarketReadyAt + at], record / kMarketSlots + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        has_pixel = false, st.alive = false, pend.shadow = pend.finish = false;
                        q = n_work;
                        retired = true;
                        atomicAdd(&compact_count[4], 1u);
                    }
                }
            }
            if (lanes_where(!retired) == 0)
            {
                __builtin_amdgcn_s_sleep(64); // an emptied wavefront: nothing to issue until the next event
                continue;
            }
            if (retired)
            {
                if (!C::kPool)
                    continue;
                helper = true;
            }
        }
        if (!helper && !has_pixel && q >= n_work)
        {
            if (!C::kPool)
                break;
            helper = true;
        }
        if (C::kPool && !(kEventsBuilt && events_on) && __ballot(!helper) == 0)
            break; // every lane of the wavefront that is here is out of work
        if (!helper && !has_pixel)
        {
            // (n_work is a multiple of 64: whole tiles)
            const uint32_t qs = job.scatter ? (q & 63u) * (n_work >> 6) + (q >> 6) : q;
            const uint32_t k = split == 1 ? 0u : qs / job.n_items, position = qs - k * job.n_items;
            // hand-out position -> item (RenderJob::tile_order: most expensive tiles first) -> tile -> pixel
            const uint32_t item = job.tile_order ? (static_cast<uint32_t>(job.tile_order[position >> 6]) << 6) | (position & 63u) : position;
            const uint32_t local_tile = item >> 6, r = item & 63u;
            const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
            const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
            my_tile = local_tile, steps = 0;
            my_lg = !levels ? 0u : position < job.level_until[0] ? 3u : position < job.level_until[1] ? 2u : position < job.level_until[2] ? 1u : 0u;
            // the lane's NEXT item: whatever the work counter hands out when the lane is free again (the launch's lanes start on
            // items 0 .. stride-1), or, without a counter, the next of its fixed list
            q = job.work_counter ? kFetchNext : q + stride;
            if (x >= width || y >= height)
            {
                finished_item = true;
                continue; // padding of an edge tile
            }
            const uint32_t pixel = y * width + x;
            if (job.wave_clock)
            {
                // (diagnostic: when this wavefront last took a pixel, and how many it took)
                unsigned long long *wc = job.wave_clock + 4u * (blockIdx.x * (kBlockSize / 64u) + (threadIdx.x >> 6));
                wc[2] = wall_clock64();
                atomicAdd(&wc[3], 1ull);
            }
            start_pixel(st, pixel);
            st.sample = k;
            slot = (job.packed ? item : pixel) + k * job.plane_stride;
            has_pixel = true;
        }
        // (merged queries: a pixel whose last sample still waits for its last shadow ray is written one step later)
        if (!helper && !st.alive && !(C::kPoolDual && pend.shadow && st.sample >= sc.camera.spp))
        {
            if (st.sample >= sc.camera.spp)
            {
                const V3 c = split == 1 ? pixel_value(sc, st) : st.pixel_sum;
                float *dst = out + 3 * static_cast<size_t>(slot);
                dst[0] = c.x, dst[1] = c.y, dst[2] = c.z;
                if (job.tile_steps)
                    atomicAdd(&job.tile_steps[my_tile], steps);
                has_pixel = false;
                finished_item = true;
                continue;
            }
            start_sample(sc, st, split, independent, job.rng_seed);
            if (kCount)
                ++local.samples;
        }
        phase_mark(kPhaseRegenerate, !helper);
        if constexpr (C::kPoolDual)
            path_step_merged<C>(sc, st, pend, cnt, !helper && st.alive, SampleStart{split, independent, job.rng_seed});
        else if constexpr (C::kPool)
            path_step_uniform<C>(sc, st, cnt, !helper, SampleStart{split, independent, job.rng_seed});
        else
            path_step<C>(sc, st, cnt);
        steps += helper ? 0u : 1u;
    }

    if (job.wave_clock && (threadIdx.x & 63u) == 0)
        job.wave_clock[4u * (blockIdx.x * (kBlockSize / 64u) + (threadIdx.x >> 6)) + 1u] = wall_clock64();
#if MCPT_PHASE_CLOCK
    if (job.phase_sums && (threadIdx.x & 63u) == 0)
    {
        const unsigned long long *a = phase_area();
        for (uint32_t k = 1; k < 1 + 3 * kPhaseCount; ++k)
            atomicAdd(job.phase_sums + k, a[k]);
    }
#endif
    if (kCount)
    {
        atomicAdd(&counters->closest_rays, static_cast<unsigned long long>(local.closest_rays));
        atomicAdd(&counters->shadow_rays, static_cast<unsigned long long>(local.shadow_rays));
        atomicAdd(&counters->node_tests, static_cast<unsigned long long>(local.node_tests));
        atomicAdd(&counters->prim_tests, static_cast<unsigned long long>(local.prim_tests));
        atomicAdd(&counters->shaded_hits, static_cast<unsigned long long>(local.shaded_hits));
        atomicAdd(&counters->samples, static_cast<unsigned long long>(local.samples));
        if (local.wave_node_steps)
            atomicAdd(&counters->wave_node_steps, static_cast<unsigned long long>(local.wave_node_steps));
        if (local.wave_prim_steps)
            atomicAdd(&counters->wave_prim_steps, static_cast<unsigned long long>(local.wave_prim_steps));
    }
}

#if !defined(MCPT_WAVE_EMU)
// THE RECORDS BEHIND A POINTER (render_kernel.h, LaunchRecords; EXPERIMENTS R6-8): which instantiations read the scene and the job
// through a pointer to device memory instead of taking them by value.  Measured per unit (same box, frames identical): the lean
// LDS-resident pool-walk kernels (cornell 37.6 -> 37.0 ms), the one-BSDF surface units (matpreview 415 -> 412, 672 -> 668) and the
// class-sorted kernels (volumetric-caustic 649 -> 637); dragon's unit moves inside its own +-13 % in both directions and keeps the
// arguments.  -DMCPT_SCENE_POINTER=0 / 1 (on every unit) builds all of them one way for A/B measurements.
template <uint32_t kFeatures, bool kLdsGeometry>
constexpr bool records_behind_pointer()
{
#if defined(MCPT_SCENE_POINTER)
    return MCPT_SCENE_POINTER != 0;
#else
    return (kLdsGeometry && (kFeatures & kFeatPoolWalk) != 0 && (kFeatures & (kFeatVolPath | kFeatAnalytic | kFeatMicrofacet | kFeatTextures)) == 0) ||
           (kFeatures & (kFeatConductorOnly | kFeatDielectricOnly)) != 0;
#endif
}
// (the constant address space: the loads are scalar loads, like the ones from the kernel-argument segment)
typedef const LaunchRecords __attribute__((address_space(4))) *LaunchRecordsPtr;
#define MCPT_RECORDS_SCENE(records) (*(const DeviceScene *)(&(records)->sc))
#define MCPT_RECORDS_JOB(records) (*(const RenderJob *)(&(records)->job))

template <uint32_t kFeatures, bool kCount, bool kLdsGeometry>
__global__ void __launch_bounds__(kBlockSize, (Budget<kFeatures, kLdsGeometry>::kWavesPerSimd))
render_kernel(const DeviceScene sc_in, const RenderJob job, float *__restrict__ out, TraceCounters *__restrict__ counters)
{
    render_body<kFeatures, kCount, kLdsGeometry>(sc_in, job, out, counters);
}
template <uint32_t kFeatures, bool kCount, bool kLdsGeometry>
__global__ void __launch_bounds__(kBlockSize, (Budget<kFeatures, kLdsGeometry>::kWavesPerSimd))
render_kernel(LaunchRecordsPtr records, float *__restrict__ out, TraceCounters *__restrict__ counters)
{
    render_body<kFeatures, kCount, kLdsGeometry>(MCPT_RECORDS_SCENE(records), MCPT_RECORDS_JOB(records), out, counters);
}

// The same kernel under its own name for the low-spp COST PROBE (RenderJob::tile_steps; capi.cpp, CostOrderedTable), so that a
// kernel trace tells the probe launch of a renderer's first draw from the frames (only the diffuse LDS instantiations are probed).
template <uint32_t kFeatures>
__global__ void __launch_bounds__(kBlockSize, (Budget<kFeatures, true>::kWavesPerSimd))
cost_probe_kernel(const DeviceScene sc_in, const RenderJob job, float *__restrict__ out)
{
    render_body<kFeatures, false, true>(sc_in, job, out, nullptr);
}
template <uint32_t kFeatures>
__global__ void __launch_bounds__(kBlockSize, (Budget<kFeatures, true>::kWavesPerSimd))
cost_probe_kernel(LaunchRecordsPtr records, float *__restrict__ out)
{
    render_body<kFeatures, false, true>(MCPT_RECORDS_SCENE(records), MCPT_RECORDS_JOB(records), out, nullptr);
}
#endif // !MCPT_WAVE_EMU

constexpr uint32_t kAll = kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet;
constexpr uint32_t kSurface = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
constexpr uint32_t kVolumeLean = kFeatVolPath | kFeatAnalytic | kFeatMicrofacet; // volume paths, quadrics, every BSDF; no emitters, constant textures
// kV: the walk of scenes outside LDS — vote-scheduled, on the binary hierarchy.  -DMCPT_WIDE_WALK=1 builds these
// instantiations on the 4-wide quantised hierarchy with the short stack instead (short_stack.h): exact (same goldens),
// measured, not faster inside the render kernels — dragon 98.1 -> 100.8 ms, matpreview 332 -> 335 and 515 -> 522 ms at
// a quarter of the spp (profiles/r03_experiments/wide_walk.md): half the node steps at twice the VALU work per step, and
// the walk is bound by VALU issue on diverged wavefronts, not by records fetched.
#ifndef MCPT_WIDE_WALK
#define MCPT_WIDE_WALK 0
#endif
constexpr uint32_t kP = kFeatOrderedWalk | kFeatPoolWalk; // the wavefront-cooperative pool walk (LDS-resident scenes)
constexpr uint32_t kPM = kP | kFeatPoolMerge;             // ... with merged queries (path_core.h: two ray records per lane)
constexpr uint32_t kPBU = kP | kFeatPoolBig;              // ... on scenes outside LDS, two queries per vertex
constexpr uint32_t kPB = kPBU | kFeatPoolMerge;           // ... with merged queries (the path integrator's kernels: not with kFeatVolPath)
constexpr uint32_t kO = kFeatOrderedWalk, kV = kFeatOrderedWalk | kFeatVoteWalk | (MCPT_WIDE_WALK ? kFeatWideWalk : 0u), kS = kFeatSlivers;
// stack entries per lane in LDS: the ring of the short stack, or one entry per level of the binary hierarchy
inline size_t WalkStackEntries(const DeviceScene &sc, uint32_t features)
{
    return (features & kFeatWideWalk) ? size_t(kWideRing) : size_t(sc.integrator.walk_depth);
}

inline size_t StagedBytes(const DeviceScene &sc, bool ordered, bool pool = false)
{
    size_t vecs = 2ull * sc.integrator.n_nodes + 3ull * sc.integrator.n_prims;
    if (ordered)
        vecs += (pool ? static_cast<unsigned long long>(kPoolLdsNodeVecs) * sc.integrator.n_pool_nodes : 4ull * sc.integrator.n_walk_nodes) + 3ull * sc.integrator.n_prims;
    return vecs * sizeof(float4);
}

void NoteTransposed(bool transposed);

// The launch's dynamic LDS: the staged traversal data, the wavefronts' pool areas (or the lanes' stack columns) and the compaction's
// words.  (One function for the launcher and for the lockstep host build of the kernel body, tests/emu: an area sized for another
// instantiation than the one that runs is exactly the kind of error that build exists to find.)
template <uint32_t kFeatures, bool kCount, bool kLdsGeometry>
inline size_t LaunchLdsBytes(const DeviceScene &sc)
{
    constexpr bool kOrdered = (kFeatures & kFeatOrderedWalk) != 0;
    constexpr bool kPool = (kFeatures & kFeatPoolWalk) != 0;
    static_assert(!kPool || (kOrdered && (kLdsGeometry != ((kFeatures & kFeatPoolBig) != 0))), "pool walk: 16-bit items with the hierarchy staged in LDS, 32-bit items outside");
    static_assert((kBlockSize / 64u) * pool_wave_words(false) >= kCompactWords * kBlockSize, "the compaction's words travel through the pool areas");
    return (kLdsGeometry ? StagedBytes(sc, kOrdered, kPool) : 0) +
           (kPool      ? size_t(kBlockSize / 64u) * pool_wave_words((kFeatures & kFeatAnalytic) != 0, (kFeatures & kFeatPoolBig) != 0, Config<kFeatures>::kPoolDual) * sizeof(uint32_t) + 8 * sizeof(uint32_t)
            : kOrdered ? WalkStackEntries(sc, kFeatures) * kBlockSize * sizeof(uint32_t)
                       : 0) +
           (!kPool && kLdsGeometry && !kCount && kOrdered && !(kFeatures & (kFeatVolPath | kFeatAnalytic))
                ? (size_t(kCompactWords) * kBlockSize + 8) * sizeof(uint32_t)
                : 0);
}

// What a launch on `max_blocks` CUs with `per_cu` resident workgroups each does with a job: lanes per path, pixel order, the level
// thresholds of its occupancy — and the number of workgroups (0: nothing to do).
template <uint32_t kFeatures, bool kLdsGeometry>
inline uint64_t ShapeLaunch(const RenderJob &job, int per_cu, uint32_t max_blocks, RenderJob &spread_job)
{
    constexpr bool kPool = (kFeatures & kFeatPoolWalk) != 0;
    const uint32_t n_work = job.n_items * (job.sample_split ? job.sample_split : 1u);
    const uint32_t resident = max_blocks * static_cast<uint32_t>(per_cu);
    spread_job = job;
    // (measured on rank shares of cornell-box and volumetric-caustic: this kernel's wavefronts execute nearly the same
    //  instructions with 8 paths as with 64 — dense is the default; cornell's 1/8 share gains 8 % at spread 2 - 4)
    //  — except the diffuse LDS instantiations on a quarter of the lanes or less: 1 path per 2 lanes, cornell's 1/4 and
    //  1/8 shares 43.6 -> 38.8 and 41.2 -> 38.8 ms; its 1/2 share 42.8 -> 53.0 ms, so not there)
    if (spread_job.lane_spread == 0)
    {
        spread_job.lane_spread =
            kLdsGeometry && (kFeatures & kAll & ~kFeatEmitters) == 0 && uint64_t(n_work) * 4u <= uint64_t(resident) * kBlockSize ? 2u : 1u;
        if (kPool)
        {
            // pool walk: the lanes between the paths are helpers of their wavefront's ray queries, which shortens a pixel's chain
            // of samples — as long as the SIMDs are not full: the helpers' instructions cost issue slots like anybody's.  Measured
            // on rank shares of cornell 512 x 512 spp 256 (profiles/r04_experiments/pool_walk_rank_shares.jsonl), lanes used =
            // pixels x spread: a 1/4 share 43.7 / 35.9 / 43.6 ms at a quarter / half / all of the GPU's lanes, a 1/8 share
            // 35.4 / 32.7 / 40.3 ms; a 1/2 share 46.4 ms dense on half of the lanes, 40.9 ms spread over all of them.  Rule: half
            // of the lanes — all of them when that leaves the paths dense.
            const uint64_t lanes = uint64_t(resident) * kBlockSize;
            uint32_t spread = 1;
            while (spread < kPoolMaxSpread && uint64_t(n_work) * (spread * 2u) * 2u <= lanes)
                spread *= 2u;
            if (spread == 1 && uint64_t(n_work) * 2u <= lanes)
                spread = 2;
            spread_job.lane_spread = spread;
        }
    }
    if (kPool && spread_job.lane_spread > 1)
        spread_job.compact = 0; // (the compaction would gather the paths into the first wavefronts again)
    if (per_cu >= 4)
        for (int i = 0; i < 3; ++i)
            spread_job.level_until[i] = job.level_until_4[i];
    if (spread_job.scatter == kScatterAuto)
        // (measured on the diffuse instantiations; volumetric-caustic lost 6 % with it at 3.5 pixels per lane)
        spread_job.scatter = kLdsGeometry && (kFeatures & kAll & ~kFeatEmitters) == 0 &&
                                     uint64_t(n_work) <= uint64_t(resident) * kBlockSize
                                 ? 1u
                                 : 0u;
    uint64_t blocks = (uint64_t(n_work) * spread_job.lane_spread + kBlockSize - 1) / kBlockSize;
    return blocks > resident ? resident : blocks;
}

#if !defined(MCPT_WAVE_EMU)
template <uint32_t kFeatures, bool kCount, bool kLdsGeometry = false>
hipError_t Launch(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters, hipStream_t stream,
                  uint32_t max_blocks)
{
    constexpr bool kByPointer = records_behind_pointer<kFeatures, kLdsGeometry>();
    constexpr bool kProbed = kLdsGeometry && !kCount && (kFeatures & kAll & ~kFeatEmitters) == 0; // (only the diffuse LDS instantiations are probed)
    // (the two forms of a kernel are overloads of one name: a kernel trace shows `render_kernel<...>` either way)
    using ByValue = void (*)(const DeviceScene, const RenderJob, float *, TraceCounters *);
    using ByPointer = void (*)(LaunchRecordsPtr, float *, TraceCounters *);
    using ProbeByValue = void (*)(const DeviceScene, const RenderJob, float *);
    using ProbeByPointer = void (*)(LaunchRecordsPtr, float *);
    const typename std::conditional<kByPointer, ByPointer, ByValue>::type kernel = render_kernel<kFeatures, kCount, kLdsGeometry>;
    const size_t lds_bytes = LaunchLdsBytes<kFeatures, kCount, kLdsGeometry>(sc);
    int per_cu = 0;
    hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlockSize, lds_bytes);
    if (err != hipSuccess)
        return err;
    if (per_cu < 1)
        per_cu = 1;
    RenderJob spread_job;
    const uint64_t blocks = ShapeLaunch<kFeatures, kLdsGeometry>(job, per_cu, max_blocks, spread_job);
    NoteTransposed(spread_job.scatter != 0);
    if (blocks == 0)
        return hipSuccess;
    const dim3 grid(static_cast<uint32_t>(blocks)), block(kBlockSize);
    if constexpr (kByPointer)
    {
        err = StageLaunchRecords(sc, spread_job, stream);
        if (err != hipSuccess)
            return err;
        const LaunchRecords *records = spread_job.launch_records;
        bool probe = false;
        if constexpr (kProbed)
            probe = job.tile_steps != nullptr;
        if constexpr (kProbed)
        {
            if (probe)
            {
                const ProbeByPointer probe_kernel = cost_probe_kernel<kFeatures>;
                hipLaunchKernelGGL(probe_kernel, grid, block, lds_bytes, stream, (LaunchRecordsPtr)(records), out);
            }
        }
        if (!probe)
            hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, (LaunchRecordsPtr)(records), out, counters);
        return hipGetLastError();
    }
    else
    {
        if constexpr (kProbed)
        {
            if (job.tile_steps)
            {
                const ProbeByValue probe_kernel = cost_probe_kernel<kFeatures>;
                hipLaunchKernelGGL(probe_kernel, grid, block, lds_bytes, stream, sc, spread_job, out);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, sc, spread_job, out, counters);
        return hipGetLastError();
    }
}

// who instantiates what
#define MCPT_LAUNCH_ARGS const DeviceScene &, const RenderJob &, float *, TraceCounters *, hipStream_t, uint32_t
#if !defined(MCPT_UNIT_ALL)
extern template hipError_t Launch<kAll, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kV, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kV | kS, false, false>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_COUNTED)
extern template hipError_t Launch<kAll, true, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kV | kS, true, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kPB | kS, true, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kP, true, true>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_LDS)
extern template hipError_t Launch<kAll | kO, false, true>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kVolumeLean | kO, false, true>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_LEAN_POOL)
extern template hipError_t Launch<kP, false, true>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kFeatEmitters | kP, false, true>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_LEAN_POOL_MERGED)
extern template hipError_t Launch<kPM, false, true>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kFeatEmitters | kPM, false, true>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_POOL)
extern template hipError_t Launch<kSurface | kPB, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kSurface | kPB | kS, false, false>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_POOL_2)
extern template hipError_t Launch<kFeatEmitters | kPB, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kFeatEmitters | kPB | kS, false, false>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_POOL_4)
extern template hipError_t Launch<kAll | kPB, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kPB | kS, false, false>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_POOL_3)
extern template hipError_t Launch<kSurface | kPBU | kFeatConductorOnly, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kSurface | kPBU | kFeatDielectricOnly, false, false>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_SURFACE)
extern template hipError_t Launch<kSurface | kV, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kSurface | kV | kS, false, false>(MCPT_LAUNCH_ARGS);
#endif
#endif // !MCPT_WAVE_EMU

} // namespace mcpt

#endif // MCPT_RENDER_KERNEL_IMPL_H
