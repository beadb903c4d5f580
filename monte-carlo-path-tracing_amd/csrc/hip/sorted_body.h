// The body of the class-sorted kernel (hip/sorted_kernel.hip holds the kernels' entry points and the launcher): shared with the
// lockstep host build of the wavefront-cooperative code (tests/emu/wave_emu.cpp, csrc/wave_target.h).
//
// The lane-owns-a-path kernel with a CLASS SORT between the ray query and the shading (north_star: "a persistent-threads
// shade stage that sorts hit records by material in LDS"), for the full-feature scenes whose traversal data sits in LDS.
//
// What it cures.  render_kernel's step is extend -> resolve -> roulette -> connect -> scatter for 64 lanes in lock
// step.  On a scene with a participating medium, quadrics and several BSDF models (volumetric-caustic) the walk is
// only a third of the kernel's VALU instructions; the other two thirds are shading code that runs at a quarter of the
// lanes (measured: 552 VALU wave-instructions per sample, 187 of them in the walk at 0.63 / 0.43 lane utilisation,
// 365 in the rest at 0.24), because the lanes of a wavefront sit at different kinds of vertices — medium scattering
// event, diffuse wall, glass, pass-through boundary, light, miss — and the wavefront executes every kind's code.
//
// What it does.  A workgroup of 256 lanes steps in lock step.  A step is cut where the kinds part ways: extend (the
// closest-hit query), resolve (surface frame, free-flight sampling of the medium the ray crossed, escape / light / back
// face) and the roulette run first; then every path has a class — medium scattering event, surface vertex by BSDF kind,
// finished sample (nothing left to do in this step), exhausted lane — and the paths are counting-sorted by class over the
// workgroup: per wavefront one ballot and population count per class, one LDS word per (wavefront, class), a prefix over
// those 4 x K words gives every path its destination lane, and its state (RNG, pixel bookkeeping, sums, the vertex and
// its shading frame: 36 words) moves through LDS to that lane.  Then connect (light sampling, shadow query, BSDF / phase
// evaluation, MIS) and scatter run on wavefronts that hold one or two classes instead of all of them, and the wavefronts
// at the end of the order hold only finished samples: they skip that half and start their next camera rays together.
// Which lane carries a path is irrelevant to its pixel (the state is all there is) — the frame is bit for bit
// render_kernel's.
#ifndef MCPT_SORTED_BODY_H
#define MCPT_SORTED_BODY_H

#include "render_kernel_impl.h"

namespace mcpt
{

#ifndef MCPT_SORT_PASSES
#define MCPT_SORT_PASSES 3
#endif
#ifndef MCPT_SORT_PASSES_LAZY
#define MCPT_SORT_PASSES_LAZY 2
#endif
// The exchange: 36 words in MCPT_SORT_PASSES passes — or, where the surface frame is built behind the sort (instantiations without
// textures: the raw hit travels instead of the frame), 32 words in two passes of 16: 8 KB per workgroup, four barriers a step instead of
// six (with the lean instantiation's 6.6 KB of geometry, 4.5 KB of stacks: 19.4 KB, eight workgroups per CU).
constexpr uint32_t sort_words(uint32_t features) { return (features & kFeatTextures) ? 36u : 32u; }
constexpr uint32_t sort_passes(uint32_t features) { return (features & kFeatTextures) ? MCPT_SORT_PASSES : MCPT_SORT_PASSES_LAZY; }
constexpr uint32_t sort_pass_words(uint32_t features) { return sort_words(features) / sort_passes(features); }
constexpr uint32_t kSortClasses = 10; // medium vertex, surface without BSDF, 6 BSDF kinds, finished sample, exhausted lane
constexpr uint32_t kClassIdle = 8, kClassExhausted = 9;

// Wavefronts per SIMD the instantiations are compiled for.  Round 3 measured 2 / 3 / 4 on volumetric-caustic's (lean) instantiation as
// 150.8 / 116.4 / 138.8 ms at spp 128: four spilled too much and LDS held too few workgroups.  Since round 5 the lean instantiation
// spills 18 VGPRs at four (the medium and library code got shorter: EXPERIMENTS R5-9), the stacks interleave at 128 lanes and the
// exchange runs in three passes of 12 words — 17.8 KB per workgroup, eight workgroups per CU: **800.9 -> 717.4 ms** at full size
// (R5-11).  The instantiations with every feature spill 140-270 VGPRs at four and stay at three.
#ifndef MCPT_SORTED_WAVES
#define MCPT_SORTED_WAVES 3
#endif
#ifndef MCPT_SORTED_WAVES_LEAN
#define MCPT_SORTED_WAVES_LEAN 4
#endif
constexpr uint32_t sorted_waves(uint32_t features)
{
    // (outside LDS: the one-BSDF instantiations fit 128 VGPRs like their unsorted twins, hip/render_kernel_impl.h Budget)
    return (features & kFeatPoolBig) ? ((features & (kFeatConductorOnly | kFeatDielectricOnly)) ? 4u : 3u)
           : (features & (kFeatEmitters | kFeatTextures)) == 0 ? MCPT_SORTED_WAVES_LEAN : MCPT_SORTED_WAVES;
}

// Class of a path after resolve + roulette.  Order = order of the sorted sequence.
__device__ __forceinline__ uint32_t path_class(const DeviceScene &sc, const PathState &st, const Surface &surf)
{
    if (st.in_medium)
        return 0u;
    const uint32_t bsdf = sc.instances[surf.inst].bsdf;
    if (bsdf == kNone)
        return 1u;
    const uint32_t k = sc.bsdfs[bsdf].kind; // kBsdfDiffuse = 2 ... kBsdfPlastic = 7 (a light ends the path in resolve)
    return k >= kBsdfDiffuse && k <= kBsdfPlastic ? k : 2u;
}

// Measured on volumetric-caustic (1280 x 720 spp 128, one box, unsorted kernel 120.5 ms): 256 lanes 116.7-117.8 ms, 128
// lanes 115.8-116.6 (a smaller group sorts less purely but waits for fewer wavefronts at its barriers); 1 / 2 / 3 exchange
// passes: 136.8 (LDS costs a workgroup per CU) / 116.4 / 116.2; (round 3's code) 2 / 3 / 4 wavefronts per SIMD: 150.8 / 116.4 / 138.8.
#ifndef MCPT_SORT_LANES
#define MCPT_SORT_LANES 128
#endif
constexpr uint32_t kSortLanes = MCPT_SORT_LANES; // lanes of a workgroup = paths sorted together

// The launch's dynamic LDS (one function for the launcher and the lockstep host build, tests/emu).
template <uint32_t kFeatures, bool kLdsGeometry>
inline size_t SortedLdsBytes(const DeviceScene &sc)
{
    constexpr bool kPool = (kFeatures & kFeatPoolWalk) != 0, kBig = (kFeatures & kFeatPoolBig) != 0;
    constexpr uint32_t kWaves = kSortLanes / 64u;
    const size_t counters = 2u * kWaves * 16u;
    return (kLdsGeometry ? StagedBytes(sc, true, kPool) : 0) +
           (kPool ? size_t(kWaves) * pool_wave_words((kFeatures & kFeatAnalytic) != 0, kBig) + counters
                  : size_t(sc.integrator.walk_depth) * kSortLanes + // (the stacks: one column per lane of the workgroup)
                        size_t(sort_pass_words(kFeatures)) * kSortLanes + counters) * sizeof(uint32_t);
}

// kLdsGeometry: the traversal data is staged in LDS (volumetric-caustic's class).  false (round 6): scenes OUTSIDE LDS — the ray
// queries are the pool walk with 32-bit items on the quantised hierarchy (kFeatPoolBig), geometry comes through the caches, and the
// exchange travels through the wavefronts' pool areas: the surface-material meshes (matpreview: BASELINE's "BSDF-sort path").
template <uint32_t kFeatures, bool kLdsGeometry>
__device__ __forceinline__ void sorted_body(const DeviceScene &sc_in, const RenderJob &job, float *__restrict__ out)
{
    using C = Config<kFeatures>;
    static_assert(!MCPT_WAVE_DEVICE || C::kStackStride == kSortLanes, "the class-sorted kernels' stack columns interleave at their workgroup size (kFeatGroup128)");
    static_assert(kLdsGeometry || (C::kPool && C::kPoolBig), "outside LDS the ray queries are the pool walk with 32-bit items");
    static_assert(!C::kPoolDual, "the class-sorted kernels query twice per vertex: the sort sits between the two queries");
    constexpr uint32_t kBlockSize = kSortLanes; // (shadows mcpt::kBlockSize)
    MCPT_DYNAMIC_LDS(float4, lds_geometry);
    // (diagnostic, RenderJob::wave_clock: four words per wavefront — start, end, last pixel taken, pixels taken)
    unsigned long long *const wave_clock = job.wave_clock ? job.wave_clock + 4u * (blockIdx.x * (kBlockSize / 64u) + (threadIdx.x >> 6)) : nullptr;
    if (wave_clock && (threadIdx.x & 63u) == 0)
        wave_clock[0] = wall_clock64();
#if MCPT_PHASE_CLOCK
    if ((threadIdx.x & 63u) == 0)
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
        // (pool walk, pool_walk.h: the 4-wide exact form of the hierarchy instead of the binary one)
        const uint32_t n_walk_vec = C::kPool ? kPoolLdsNodeVecs * sc_in.integrator.n_pool_nodes : 4u * sc_in.integrator.n_walk_nodes, n_slot_vec = n_tri_vec;
        for (uint32_t i = threadIdx.x; i < n_node_vec; i += blockDim.x)
            lds_geometry[i] = sc_in.nodes[i];
        for (uint32_t i = threadIdx.x; i < n_tri_vec; i += blockDim.x)
            lds_geometry[n_node_vec + i] = sc_in.tri_pos[i];
        // (pool walk: the staged node records lie kPoolLdsNodeVecs vectors apart — LDS banks, pool_walk.h)
        for (uint32_t i = threadIdx.x; i < (C::kPool ? 8u * sc_in.integrator.n_pool_nodes : n_walk_vec); i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + (C::kPool ? (i >> 3) * kPoolLdsNodeVecs + (i & 7u) : i)] = C::kPool ? sc_in.pool_nodes[i] : sc_in.walk_nodes[i];
        for (uint32_t i = threadIdx.x; i < n_slot_vec; i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + n_walk_vec + i] = sc_in.walk_prims[i];
        sc.nodes = lds_geometry;
        sc.tri_pos = lds_geometry + n_node_vec;
        if (C::kPool) // (whichever form was staged; the binary form stays in HBM for the rare ray walked the per-lane way)
            sc.pool_nodes = lds_geometry + n_node_vec + n_tri_vec;
        else
            sc.walk_nodes = lds_geometry + n_node_vec + n_tri_vec;
        sc.walk_prims = lds_geometry + n_node_vec + n_tri_vec + n_walk_vec;
        n_staged = n_node_vec + n_tri_vec + n_walk_vec + n_slot_vec;
    }
    constexpr uint32_t kWaves = kBlockSize / 64u;
    constexpr uint32_t kSortWords = sort_words(kFeatures), kSortPasses = sort_passes(kFeatures), kSortPassWords = sort_pass_words(kFeatures);
    uint32_t *lds_words = reinterpret_cast<uint32_t *>(lds_geometry + n_staged);
    // one walk per lane: the lanes' stack columns, then the exchange words.  Pool walk: one pool area per wavefront, and the
    // exchange words travel THROUGH the pool areas (no wavefront is inside a query between the count barrier and the barrier
    // behind the last exchange read), the counters behind them.
    static_assert(!C::kPool || kWaves * pool_wave_words(C::kAnalytic, C::kPoolBig) >= kSortPassWords * kBlockSize, "the exchange fits the pool areas");
    uint32_t *stack = C::kPool ? lds_words + (threadIdx.x >> 6) * pool_wave_words(C::kAnalytic, C::kPoolBig) : lds_words + threadIdx.x;
    if (!C::kPool)
        lds_words += static_cast<size_t>(sc_in.integrator.walk_depth) * kSortLanes;
    uint32_t *exchange = lds_words;                                  // kSortPassWords x kSortLanes words, word-major
    uint32_t *counts = C::kPool ? lds_words + kWaves * pool_wave_words(C::kAnalytic, C::kPoolBig) : exchange + kSortPassWords * kBlockSize; // [parity][wavefront][class]
    __syncthreads(); // geometry staged

    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);
    const uint32_t split = job.sample_split ? job.sample_split : 1u, n_work = job.n_items * split;
    const bool independent = job.independent_samples != 0;
    const uint32_t wave = threadIdx.x >> 6;

    PathState st{};
    st.alive = false;
    st.stack = stack;
    bool has_pixel = false;
    uint32_t slot = 0;
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;

    for (uint32_t step = 0;; ++step)
    {
        MCPT_WAVE_CONVERGE();
        // ---- every lane that can still get work holds a live path ----
        while (!st.alive)
        {
            if (!has_pixel)
            {
                // (the next item is taken from the work counter when the lane is free, not reserved ahead: render_kernel_impl.h)
                if (job.work_counter && q == kFetchNext)
                {
                    MCPT_WAVE_REGION();
                    q = stride + wave_reserve(job.work_counter, true);
                }
                if (q >= n_work)
                    break; // exhausted
                const uint32_t qs = job.scatter ? (q & 63u) * (n_work >> 6) + (q >> 6) : q;
                const uint32_t k = split == 1 ? 0u : qs / job.n_items, position = qs - k * job.n_items;
                const uint32_t item = job.tile_order ? (static_cast<uint32_t>(job.tile_order[position >> 6]) << 6) | (position & 63u) : position;
                const uint32_t local_tile = item >> 6, r = item & 63u;
                const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
                const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
                q = job.work_counter ? kFetchNext : q + stride;
                if (x >= width || y >= height)
                    continue; // padding of an edge tile
                const uint32_t pixel = y * width + x;
                if (wave_clock)
                    wave_clock[2] = wall_clock64(), atomicAdd(&wave_clock[3], 1ull);
                start_pixel(st, pixel);
                st.sample = k;
                slot = (job.packed ? item : pixel) + k * job.plane_stride;
                has_pixel = true;
            }
            if (st.sample >= sc.camera.spp)
            {
                const V3 c = split == 1 ? pixel_value(sc, st) : st.pixel_sum;
                float *dst = out + 3 * static_cast<size_t>(slot);
                dst[0] = c.x, dst[1] = c.y, dst[2] = c.z;
                has_pixel = false;
                continue;
            }
            start_sample(sc, st, split, independent, job.rng_seed);
        }

        phase_mark(kPhaseRegenerate);
        // ---- extend, resolve, roulette ----
        // Instantiations without textures build the surface record in two halves (traversal.h, make_surface_part): resolve gets
        // position and shading normal, the raw hit travels through the exchange in the tangent frame's words, and the frame is built
        // behind the sort for the paths that go on from a surface vertex — 45 % of volumetric-caustic's vertices are medium vertices
        // that never use one, and a quadric's frame (two inverse trigonometric functions, four sinf / cosf) ran for 4 lanes of a
        // wavefront in 9 of 10 steps while the lanes of its BSDF kind were spread over the workgroup (EXPERIMENTS R5-9, R5-12).
        constexpr bool kLazyFrame = !C::kTextures;
        HitRaw hit;
        hit.inst = hit.prim = 0, hit.a = hit.b = hit.c = 0.0f, hit.inside = false;
        Surface surf;
        surf.inside = false, surf.inst = 0, surf.uv = V2{0, 0};
        surf.position = surf.normal = surf.tangent = surf.bitangent = V3{0, 0, 0};
        if constexpr (C::kPool)
        {
            // (every lane makes the query call; the ones without a path work on the others' rays)
            Ray ray;
            const bool hit_valid = path_extend_uniform<C>(sc, st, nullptr, st.alive, ray, hit);
            if (st.alive)
                path_resolve<C, kLazyFrame>(sc, st, nullptr, ray, hit, hit_valid, surf);
        }
        else if (st.alive)
        {
            Ray ray;
            const bool hit_valid = path_extend<C>(sc, st, nullptr, ray, hit);
            phase_mark(kPhaseExtend);
            path_resolve<C, kLazyFrame>(sc, st, nullptr, ray, hit, hit_valid, surf);
            phase_mark(kPhaseResolve);
        }

        // ---- class sort over the workgroup ----
        const uint32_t key = st.alive ? path_class(sc, st, surf) : (has_pixel || q < n_work) ? kClassIdle : kClassExhausted;
        uint32_t *cnt = counts + (step & 1u) * kWaves * 16u;
        uint32_t rank = 0, mine = 0; // this lane's rank among its wavefront's lanes of the same class; lane c: that class's count
#pragma unroll
        for (uint32_t c = 0; c < kSortClasses; ++c)
        {
            const unsigned long long mask = __ballot(key == c);
            const uint32_t n = static_cast<uint32_t>(__popcll(mask));
            if (key == c)
                rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
            if ((threadIdx.x & 63u) == c)
                mine = n;
        }
        if ((threadIdx.x & 63u) < 16u)
            cnt[wave * 16u + (threadIdx.x & 63u)] = (threadIdx.x & 63u) < kSortClasses ? mine : 0u;
        __syncthreads();
        // destination = (paths of smaller classes) + (paths of this class in earlier wavefronts) + rank
        uint32_t before = 0, exhausted = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w)
        {
            const uint4 *row = reinterpret_cast<const uint4 *>(cnt + w * 16u);
            const uint4 r0 = row[0], r1 = row[1], r2 = row[2];
            const uint32_t n[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
            for (uint32_t c = 0; c < kSortClasses; ++c)
                before += (c < key || (c == key && w < wave)) ? n[c] : 0u;
            exhausted += n[kClassExhausted];
        }
        if (exhausted == kBlockSize)
            break; // every lane of the workgroup is out of work (uniform)
        // The order is rotated by whole wavefronts, differently per workgroup and step: the wavefronts of a workgroup sit on
        // different SIMDs, and the front of the order (medium vertices, the common and expensive class) must not always
        // land on the same one while the SIMD of the last wavefront (finished samples) idles.
#ifndef MCPT_SORT_ROTATE
#define MCPT_SORT_ROTATE 1
#endif
        const uint32_t dst = MCPT_SORT_ROTATE ? (before + rank + 64u * ((blockIdx.x + step) & (kWaves - 1u))) & (kBlockSize - 1u) : before + rank;

        const V3 vertex = st.in_medium ? st.origin : surf.position;
        uint32_t in[kSortWords], got[kSortWords];
        in[0] = st.rng, in[1] = st.pixel, in[2] = st.sample, in[3] = st.depth;
        in[4] = (st.alive ? 1u : 0u) | (st.in_medium ? 4u : 0u) | (has_pixel ? 8u : 0u) | (surf.inside ? 32u : 0u);
        in[5] = st.medium, in[6] = surf.inst;
        auto put = [&](uint32_t at, V3 v) { in[at] = as_uint(v.x), in[at + 1] = as_uint(v.y), in[at + 2] = as_uint(v.z); };
        put(7, vertex), put(10, st.wo), put(13, st.throughput), put(16, st.L), put(19, st.pixel_sum);
        put(22, surf.normal);
        if constexpr (kLazyFrame)
            in[25] = as_uint(hit.a), in[26] = as_uint(hit.b), in[27] = as_uint(hit.c), in[28] = hit.prim;
        else
            put(25, surf.tangent), put(28, surf.bitangent), in[31] = as_uint(surf.uv.u), in[32] = as_uint(surf.uv.v);
        if constexpr (kLazyFrame)
            in[29] = slot, in[30] = q, in[31] = 0;
        else
            in[33] = slot, in[34] = q, in[35] = 0;
#pragma unroll
        for (uint32_t pass = 0; pass < kSortPasses; ++pass)
        {
            if (pass != 0)
                __syncthreads(); // the previous pass's words have been read
#pragma unroll
            for (uint32_t k = 0; k < kSortPassWords; ++k)
                exchange[k * kBlockSize + dst] = in[pass * kSortPassWords + k];
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < kSortPassWords; ++k)
                got[pass * kSortPassWords + k] = exchange[k * kBlockSize + threadIdx.x];
        }
        // (the next step's first exchange write comes after that step's count barrier: every lane has read by then)
        st.rng = got[0], st.pixel = got[1], st.sample = got[2], st.depth = got[3];
        st.alive = (got[4] & 1u) != 0, st.primary = false, st.in_medium = (got[4] & 4u) != 0;
        has_pixel = (got[4] & 8u) != 0, surf.inside = (got[4] & 32u) != 0;
        st.medium = got[5], surf.inst = got[6];
        auto get = [&](uint32_t at) { return V3{as_float(got[at]), as_float(got[at + 1]), as_float(got[at + 2])}; };
        st.origin = surf.position = get(7);
        st.wo = get(10), st.throughput = get(13), st.L = get(16), st.pixel_sum = get(19);
        surf.normal = get(22);
        phase_mark(kPhaseSort, st.alive); // (the lanes counted are the paths that go on to connect and scatter)
        if constexpr (kLazyFrame)
        {
            surf.tangent = surf.bitangent = V3{0, 0, 0}, surf.uv = V2{0, 0};
            if (st.alive && !st.in_medium)
            {
                HitRaw moved;
                moved.inst = surf.inst, moved.inside = surf.inside, moved.prim = got[28];
                moved.a = as_float(got[25]), moved.b = as_float(got[26]), moved.c = as_float(got[27]);
                make_surface_part<C::kAnalytic, false, 2>(sc, moved, surf);
            }
        }
        else
        {
            surf.tangent = get(25), surf.bitangent = get(28);
            surf.uv = V2{as_float(got[31]), as_float(got[32])};
        }
        slot = got[kLazyFrame ? 29 : 33], q = got[kLazyFrame ? 30 : 34];

        // ---- connect, scatter ----
        if constexpr (C::kPool)
        {
            __syncthreads(); // every lane has read its words: the pool areas are the queries' again
            path_connect_scatter_uniform<C>(sc, st, nullptr, surf, st.alive);
        }
        else if (st.alive)
        {
            path_connect_scatter<C>(sc, st, nullptr, surf);
            phase_mark(kPhaseScatter);
        }
    }
    if (wave_clock && (threadIdx.x & 63u) == 0)
        wave_clock[1] = wall_clock64();
#if MCPT_PHASE_CLOCK
    // (the sums of all wavefronts: RenderJob::phase_sums, 64 words of their own behind the per-wavefront words)
    if (job.phase_sums && (threadIdx.x & 63u) == 0)
    {
        const unsigned long long *a = phase_area();
        for (uint32_t k = 1; k < 1 + 3 * kPhaseCount; ++k)
            atomicAdd(job.phase_sums + k, a[k]);
    }
#endif
}


} // namespace mcpt

#endif // MCPT_SORTED_BODY_H
