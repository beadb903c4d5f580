// HIP stream kernel for gfx950 (MI355X): the workgroup-local ray pool of stream_core.h.
//
// A workgroup of 256 lanes owns P path slots (P = 512 ... 1024, a launch parameter) and alternates, between
// workgroup barriers, a SHADE phase (lane t handles slots t, t + 256, ...: regenerate / next vertex / emit rays,
// emitted rays compacted into the round's ray list by wavefront ballot + prefix count) and a TRACE phase (all
// lanes drain the ray list, fetching a new ray whenever enough lanes of a wavefront are free).  The kernel is
// persistent: the grid is sized to the machine and a slot whose pixel is finished takes the next pixel of its
// sequence until the job's items are exhausted.
//
// Memory.
//   LDS   [traversal data, if it fits (kLdsGeometry)] [traversal stacks: walk_depth x 256 words, lane-interleaved]
//         [hot slot fields (rays in / hits out), if kHotInLds] [ray list: (1 + S) P words] [control words]
//   HBM   the workgroup's region of the scratch buffer: cold slot fields (path state; read and written once per
//         round by the slot's shading lane, coalesced), and the hot fields when they do not fit LDS next to deep
//         traversal stacks (meshes).  The regions of all resident workgroups are a few tens of MB: L2 / MALL
//         resident.  The frame is written once per finished pixel (12 B).
// No MFMA (no matrix-shaped work on this path).
#ifndef MCPT_STREAM_KERNEL_IMPL_H
#define MCPT_STREAM_KERNEL_IMPL_H

#include <hip/hip_runtime.h>

#include "../stream_core.h"
#include "render_kernel.h"

namespace mcpt
{

struct StreamControl // LDS, double-buffered by round parity
{
    uint32_t n_ext[2], n_shadow[2], next[2];
    uint32_t pad[2];
};

// Wavefronts per SIMD the register allocation aims for (-D overrides for experiments).  Measured on the mesh
// instantiations, 4 / 5 / 6: dragon stand-in 14.6 / 14.5 / 13.2-13.8 ms, matpreview rough conductor 38.0 / 40.6 /
// 40.0-41.2, rough dielectric 45.9 / 48.8 / 49.6-50.2 (reduced films): flat, 4 is never far from the best.  The
// full-feature small-scene instantiation: 27.0 ms at 3 against 29.9 at 4 (volumetric-caustic, reduced film).
#ifndef MCPT_STREAM_WAVES_MESH
#define MCPT_STREAM_WAVES_MESH 4
#endif
#ifndef MCPT_STREAM_WAVES_SMALL_FULL
#define MCPT_STREAM_WAVES_SMALL_FULL 3
#endif
#ifndef MCPT_STREAM_WAVES_MEMORY
#define MCPT_STREAM_WAVES_MEMORY 2
#endif
// One slot per lane (kRegs) with StreamLaunch::wave_local: every wavefront of the workgroup runs its rounds on its own
// — its own ray list and control words, no workgroup barrier in the round loop.  The shadow rays of a wavefront then
// fill only that wavefront's free lanes, and no wavefront waits for the longest ray of the other three.  Which of the
// two is faster depends on the scene (full size, Msamples/s, workgroup rounds -> wavefront rounds: matpreview 444 -> 489
// and 315 -> 346, classroom 125 -> 137, dining-room 63 -> 77, dragon/scene.xml 920 -> 794): the renderer's calibration
// times both.
constexpr uint32_t kStreamWaves = kBlockSize / 64u;
template <uint32_t kFeatures, bool kLdsGeometry, bool kRegs>
struct StreamBudget
{
    // Mesh instantiations exist at 4, 3 and 2 wavefronts per SIMD (kFeatWaves3 / kFeatWaves2).  Round 3, with wavefront rounds,
    // pre-pass and work counter in place, full configurations on one box: matpreview rough conductor 990 / 848 / 895 ms at
    // 4 / 3 / 2, rough dielectric 1268 / 1287 / 1530, dragon/scene.xml 186 / 176 / 170 — a throughput-bound frame wants the
    // wavefronts, a chain-bound one (fewer expensive pixels than lanes: dragon) wants its rounds free of spill traffic.
    static constexpr int kWavesPerSimd = !kRegs ? MCPT_STREAM_WAVES_MEMORY
                                         : (kFeatures & kFeatWaves2) ? 2
                                         : (kFeatures & kFeatWaves3) ? 3
                                         : !kLdsGeometry ? MCPT_STREAM_WAVES_MESH
                                         : (kFeatures & (kFeatVolPath | kFeatAnalytic | kFeatMicrofacet)) ? MCPT_STREAM_WAVES_SMALL_FULL
                                                                                                        : 4;
};

// kRegs: one slot per lane (P = 256) whose path state stays in the lane's registers between rounds — only rays
// and hits go through the pool.  Otherwise P slots per workgroup with their state in the scratch buffer.
template <uint32_t kFeatures, uint32_t S, bool kCount, bool kLdsGeometry, bool kHotInLds, bool kRegs>
__global__ void __launch_bounds__(kBlockSize, (StreamBudget<kFeatures, kLdsGeometry, kRegs>::kWavesPerSimd))
stream_kernel(const DeviceScene sc_in, const RenderJob job, float *__restrict__ out, TraceCounters *__restrict__ counters,
              uint32_t *__restrict__ scratch, const StreamLaunch cfg)
{
    using C = Config<kFeatures>;
    const bool kWaveLocal = kRegs && cfg.wave_local != 0; // (uniform over the launch)
    const uint32_t wave = kWaveLocal ? threadIdx.x >> 6 : 0u;
    extern __shared__ float4 lds_geometry[];
    DeviceScene sc = sc_in;
    uint32_t n_staged = 0;
    if (kLdsGeometry)
    {
        const uint32_t n_node_vec = 2u * sc_in.integrator.n_nodes, n_tri_vec = 3u * sc_in.integrator.n_prims;
        const uint32_t n_walk_vec = 4u * sc_in.integrator.n_walk_nodes, n_slot_vec = n_tri_vec;
        for (uint32_t i = threadIdx.x; i < n_node_vec; i += blockDim.x)
            lds_geometry[i] = sc_in.nodes[i];
        for (uint32_t i = threadIdx.x; i < n_tri_vec; i += blockDim.x)
            lds_geometry[n_node_vec + i] = sc_in.tri_pos[i];
        for (uint32_t i = threadIdx.x; i < n_walk_vec; i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + i] = sc_in.walk_nodes[i];
        for (uint32_t i = threadIdx.x; i < n_slot_vec; i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + n_walk_vec + i] = sc_in.walk_prims[i];
        sc.nodes = lds_geometry;
        sc.tri_pos = lds_geometry + n_node_vec;
        sc.walk_nodes = lds_geometry + n_node_vec + n_tri_vec;
        sc.walk_prims = lds_geometry + n_node_vec + n_tri_vec + n_walk_vec;
        n_staged = n_node_vec + n_tri_vec + n_walk_vec + n_slot_vec;
    }
    const uint32_t P = cfg.slots;
    uint32_t *lds_words = reinterpret_cast<uint32_t *>(lds_geometry + n_staged);
    uint32_t *stack = lds_words + threadIdx.x;
    lds_words += sc_in.integrator.walk_depth * kBlockSize;
    uint32_t *region = scratch + static_cast<size_t>(blockIdx.x) * cfg.scratch_words_per_block;
    StreamStore m;
    m.P = P;
    m.cold = region;
    if (kHotInLds)
    {
        m.hot = lds_words;
        lds_words += stream_hot_words(S) * P;
    }
    else
        m.hot = region + (kRegs ? 0u : stream_cold_words(S)) * P;
    // (wave-local: the wavefront's S x 64 shadow-ray entries sit where the workgroup's list has them, behind P)
    uint32_t *ids = lds_words + wave * S * 64u;
    StreamControl *ctrl = reinterpret_cast<StreamControl *>(lds_words + (1u + S) * P) + wave;
    if (threadIdx.x < kStreamWaves)
        reinterpret_cast<StreamControl *>(lds_words + (1u + S) * P)[threadIdx.x] = StreamControl{};
    // between the phases of a round: the workgroup's barrier, or — wave-local — only the memory order (the lanes of a
    // wavefront run in lock step; what they wrote for each other must have landed)
    auto phase_sync = [&]()
    {
        if (kWaveLocal)
        {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        else
            __syncthreads();
    };

    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);
    // independent-sample mode with split samples: item q = k * n_items + pixel item, the slot renders samples k, k + split,
    // ... of that pixel into plane k (uniform values; split == 1 in the reference mode)
    const uint32_t split = job.sample_split ? job.sample_split : 1u, n_work = job.n_items * split;
    // (one slot per lane: RenderJob::lane_spread — only every spread-th lane takes pixels)
    uint32_t spread = kRegs && cfg.lane_spread ? cfg.lane_spread : 1u;
    if (kRegs && cfg.lane_spread == 0 && job.hit_counters)
    {
        // the plan left the choice to the launch: from the pre-pass's count of camera rays that hit something
        // (expensive pixels = hits / spp, at most the job's pixels)
        unsigned long long hits = 0;
        for (uint32_t k = 0; k < kHitCounters; ++k)
            hits += job.hit_counters[k];
        unsigned long long expensive = hits / sc_in.camera.spp;
        expensive = (expensive > job.n_items ? job.n_items : expensive < 1 ? 1 : expensive) * split;
        const unsigned long long lanes = static_cast<unsigned long long>(gridDim.x) * P;
        while (spread < kMaxStreamSpread && 2ull * spread * expensive * kSpreadDen <= lanes * kSpreadNum)
            spread *= 2;
    }
    const uint32_t stride = gridDim.x * P / spread;
    // item -> tile -> pixel; false for the padding of an edge tile
    auto pixel_of = [&](uint32_t q, uint32_t &pixel) -> bool
    {
        const uint32_t local_tile = q >> 6, r = q & 63u;
        const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
        const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
        pixel = y * width + x;
        return x < width && y < height;
    };
    // gives the slot the first pixel of its sequence at or after item q (fixed lists q, q + stride, ...), or — with the
    // job's work counter — the first item nobody has taken yet (all lanes of the wavefront call this together)
    auto assign = [&](StreamSlot<S> &s, uint32_t q)
    {
        for (;; q = job.work_counter ? stride + wave_reserve(job.work_counter, true) : q + stride)
        {
            if (q >= n_work)
            {
                s.flags |= kSlotExhausted;
                return;
            }
            const uint32_t k = split == 1 ? 0u : q / job.n_items, position = q - k * job.n_items;
            // (RenderJob::tile_order, work counter only: hand-out position -> item, most expensive tiles first)
            const uint32_t item = job.tile_order ? (static_cast<uint32_t>(job.tile_order[position >> 6]) << 6) | (position & 63u) : position;
            uint32_t pixel;
            if (pixel_of(item, pixel))
            {
                s.item = item + k * job.n_items;
                start_pixel(s.st, pixel);
                s.st.sample = k;
                return;
            }
        }
    };

    LaneCounters local{};
    LaneCounters *cnt = kCount ? &local : nullptr;

    StreamSlot<S> mine{}; // kRegs: this lane's slot
    if (kRegs)
    {
        mine.flags = 0;
        if (threadIdx.x % spread != 0)
            mine.flags |= kSlotExhausted;
        else
            assign(mine, (blockIdx.x * P + threadIdx.x) / spread);
    }
    else
        for (uint32_t i = threadIdx.x; i < P; i += kBlockSize)
        {
            StreamSlot<S> s{};
            s.flags = 0;
            assign(s, blockIdx.x * P + i);
            stream_save<C, S>(m, i, s);
        }
    __syncthreads();

    unsigned long long t_shade = 0, t_trace = 0, t_wait = 0, n_rounds = 0; // kCount: shader-clock ticks of this wavefront
    auto now = [&]() -> unsigned long long { return kCount ? static_cast<unsigned long long>(clock64()) : 0ull; };
    for (uint32_t round = 0;; ++round)
    {
        const uint32_t p = round & 1u;
        const unsigned long long t0 = now();
        // ---- shade -------------------------------------------------------------------------------
        // one slot: results in, next vertex, rays out, emitted rays compacted into the round's list — extension
        // rays from the front, shadow rays behind them (wavefront ballot + prefix count, one LDS atomic per
        // wavefront and kind)
        auto shade_slot = [&](StreamSlot<S> &s, uint32_t i)
        {
            while (stream_shade<C, S>(sc, s, cnt, job.independent_samples != 0, job.rng_seed, split, job.n_items) == kStreamPixelDone)
            {
                const uint32_t k = split == 1 ? 0u : s.item / job.n_items;
                const V3 c = split == 1 ? pixel_value(sc, s.st) : s.st.pixel_sum; // (planes hold unnormalised partial sums)
                float *dst = out + 3 * (static_cast<size_t>(job.packed ? s.item - k * job.n_items : s.st.pixel) +
                                        static_cast<size_t>(k) * job.plane_stride);
                dst[0] = c.x, dst[1] = c.y, dst[2] = c.z;
                assign(s, job.work_counter ? stride + wave_reserve(job.work_counter, true) : s.item + stride);
            }
        };
        auto list_rays = [&](const StreamSlot<S> &s, uint32_t i)
        {
            const bool ext = (s.flags & kSlotExtRay) != 0;
            const uint32_t at = wave_reserve(&ctrl->n_ext[p], ext);
            if (ext)
                ids[at] = i;
#pragma unroll
            for (uint32_t k = 0; k < S; ++k)
            {
                const bool sh = (s.flags & (kSlotShadow0 << k)) != 0;
                const uint32_t at_sh = wave_reserve(&ctrl->n_shadow[p], sh);
                if (sh)
                    ids[P + at_sh] = (1u + k) * P + i;
            }
        };
        OwnRay own;
        own.active = false;
        if (kRegs)
        {
            // the lane traced its own extension ray (its answer is already in `mine`); its shadow rays went
            // through the pool
            const uint32_t ext_bit = mine.flags & kSlotExtRay;
            mine.flags &= ~kSlotExtRay;
            const bool hv = mine.hit_valid;
            const HitRaw hr = mine.hit;
            const float ht = mine.hit_t;
            stream_load_hot<C, S>(m, threadIdx.x, mine); // occlusion answers
            mine.hit_valid = hv, mine.hit = hr, mine.hit_t = ht;
            mine.flags |= ext_bit;
            shade_slot(mine, threadIdx.x);
            own.active = (mine.flags & kSlotExtRay) != 0;
            own.origin = mine.st.origin, own.dir = mine.st.dir;
            StreamSlot<S> shadows_only = mine;
            shadows_only.flags &= ~kSlotExtRay;
            stream_save_hot<C, S>(m, threadIdx.x, shadows_only);
            list_rays(shadows_only, threadIdx.x);
        }
        else
            for (uint32_t i = threadIdx.x; i < P; i += kBlockSize)
            {
                StreamSlot<S> s;
                stream_load<C, S>(m, i, s);
                shade_slot(s, i);
                stream_save<C, S>(m, i, s);
                list_rays(s, i);
            }
        const unsigned long long t1 = now();
        phase_sync();
        const unsigned long long t2 = now();
        const uint32_t n_ext = ctrl->n_ext[p], n_shadow = ctrl->n_shadow[p];
        if (kRegs)
        {
            // (with the extension rays in registers the ray list only says whether shadow rays exist: the
            //  workgroup is done when no lane has a ray of either kind)
            if (kWaveLocal)
            {
                if (lanes_where(own.active) == 0 && n_ext + n_shadow == 0)
                    break;
            }
            else if (__syncthreads_or(own.active ? 1 : 0) == 0 && n_ext + n_shadow == 0)
                break;
        }
        else if (n_ext + n_shadow == 0)
            break; // every slot is exhausted
        if (kWaveLocal ? (threadIdx.x & 63u) == 0 : threadIdx.x == 0)
            ctrl->n_ext[p ^ 1u] = 0, ctrl->n_shadow[p ^ 1u] = 0, ctrl->next[p ^ 1u] = 0;
        // ---- trace -------------------------------------------------------------------------------
        const StreamRayList list{ids, n_ext, n_shadow, &ctrl->next[p]};
        stream_trace<C, kCount, !kRegs>(sc, m, list, stack, cfg.refill_at, cnt, kRegs ? &own : nullptr);
        if (kRegs)
            mine.hit_valid = own.found, mine.hit = own.hit, mine.hit_t = own.t;
        const unsigned long long t3 = now();
        phase_sync();
        if (kCount)
            t_shade += t1 - t0, t_trace += t3 - t2, t_wait += (t2 - t1) + (now() - t3), ++n_rounds;
    }

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
        if ((threadIdx.x & 63u) == 0)
        {
            atomicAdd(&counters->ticks_shade, t_shade), atomicAdd(&counters->ticks_trace, t_trace);
            atomicAdd(&counters->ticks_wait, t_wait);
            if (threadIdx.x == 0)
                atomicAdd(&counters->rounds, n_rounds);
        }
    }
}

// LDS a workgroup of the stream kernel needs.
template <uint32_t S>
inline size_t StreamLdsBytes(const DeviceScene &sc, bool lds_geometry, bool hot_in_lds, uint32_t slots)
{
    size_t vecs = 0;
    if (lds_geometry)
        vecs = 2ull * sc.integrator.n_nodes + 6ull * sc.integrator.n_prims + 4ull * sc.integrator.n_walk_nodes;
    size_t words = size_t(sc.integrator.walk_depth) * kBlockSize + (1u + S) * size_t(slots) + kStreamWaves * sizeof(StreamControl) / 4;
    if (hot_in_lds)
        words += size_t(stream_hot_words(S)) * slots;
    return vecs * sizeof(float4) + words * sizeof(uint32_t);
}

template <uint32_t S>
inline size_t StreamScratchWordsPerBlock(bool hot_in_lds, bool regs, uint32_t slots)
{
    return size_t((regs ? 0u : stream_cold_words(S)) + (hot_in_lds ? 0u : stream_hot_words(S))) * slots;
}

// Fills in what the launch will use (grid, LDS, scratch) without launching: capi.cpp sizes the scratch buffer from it.
template <uint32_t kFeatures, uint32_t S, bool kCount, bool kLdsGeometry, bool kHotInLds, bool kRegs>
hipError_t PlanStream(const DeviceScene &sc, const RenderJob &job, uint32_t n_cus, StreamLaunch &cfg)
{
    if (kRegs)
        cfg.slots = kBlockSize;
    cfg.lds_bytes = static_cast<uint32_t>(StreamLdsBytes<S>(sc, kLdsGeometry, kHotInLds, cfg.slots));
    cfg.scratch_words_per_block = static_cast<uint32_t>(StreamScratchWordsPerBlock<S>(kHotInLds, kRegs, cfg.slots));
    int per_cu = 0;
    hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &per_cu, stream_kernel<kFeatures, S, kCount, kLdsGeometry, kHotInLds, kRegs>, kBlockSize, cfg.lds_bytes);
    if (err != hipSuccess)
        return err;
    if (per_cu < 1)
        return hipErrorOutOfMemory; // does not fit: the caller falls back to fewer slots or the other kernel
    const uint32_t resident = n_cus * static_cast<uint32_t>(per_cu);
    cfg.lane_spread = 1;
    const uint64_t n_work = uint64_t(job.n_items) * (job.sample_split ? job.sample_split : 1u);
    uint64_t blocks = (n_work + cfg.slots - 1) / cfg.slots;
    if (kRegs)
    {
        // (the automatic spread is for the scenes outside LDS: latency-bound rounds.  With the hierarchy in LDS the
        //  kernel is VALU-bound and idle lanes are not free: cornell 776 -> 519 Msamples/s at 1 path per 2 lanes)
        cfg.lane_spread = job.lane_spread ? job.lane_spread : kLdsGeometry ? 1u : 0u;
        if (cfg.lane_spread == 0 && job.hit_counters)
            blocks = resident; // the kernel sizes the spread from the pre-pass's hit count: every resident slot is launched
        else
        {
            // no pre-pass: every pixel of the job counts as expensive
            if (cfg.lane_spread == 0)
                for (cfg.lane_spread = 1; cfg.lane_spread < kMaxStreamSpread &&
                                          2ull * cfg.lane_spread * n_work * kSpreadDen <= uint64_t(resident) * cfg.slots * kSpreadNum;)
                    cfg.lane_spread *= 2;
            blocks = (n_work * cfg.lane_spread + cfg.slots - 1) / cfg.slots;
        }
    }
    cfg.blocks = blocks > resident ? resident : static_cast<uint32_t>(blocks);
    cfg.blocks_per_cu = static_cast<uint32_t>(per_cu);
    return hipSuccess;
}

template <uint32_t kFeatures, uint32_t S, bool kCount, bool kLdsGeometry, bool kHotInLds, bool kRegs>
hipError_t LaunchStream(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters, hipStream_t stream,
                        uint32_t *scratch, const StreamLaunch &cfg)
{
    if (cfg.blocks == 0)
        return hipSuccess;
    hipLaunchKernelGGL((stream_kernel<kFeatures, S, kCount, kLdsGeometry, kHotInLds, kRegs>), dim3(cfg.blocks), dim3(kBlockSize),
                       cfg.lds_bytes, stream, sc, job, out, counters, scratch, cfg);
    return hipGetLastError();
}

} // namespace mcpt

#endif // MCPT_STREAM_KERNEL_IMPL_H
