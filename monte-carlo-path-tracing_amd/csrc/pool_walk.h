// The wavefront-cooperative ray query (round 4): scenes in LDS (16-bit items) and outside (kFeatPoolBig: 32-bit items, the
// hierarchy through the caches).
//
// What it cures.  walk_ordered (traversal.h) gives a ray to a lane: a wavefront's query lasts as long as its slowest
// lane's walk (cornell: a ray needs ~10 node visits, the wavefront runs ~34 node steps and ~10 primitive phases per round:
// 29 % / 24 % of the lanes do useful work) — the reference's per-thread nested stack walk has the same shape
// (tlas.cpp:13-76, blas.cpp:18-77 inside the megakernel, renderer.cpp:88-95).  Here the rays of a wavefront's query
// (one per active lane) are RECORDS in LDS and the work is a shared LIFO of (ray, node) items and a second one of
// (ray, primitive slot) items: every step, every lane takes the next item, whichever ray it belongs to — a node step
// tests the FOUR children of a node of the 4-wide exact hierarchy (DeviceScene::pool_nodes) and pushes the ones the ray
// enters, a primitive step tests one slot.  All lanes stay busy until the lists run dry; a primitive phase runs when a
// wavefront's worth of slots waits (or nothing else is left), so both phases run full.  tests/emu's model of this
// schedule (Pool2Model) on cornell, per round (closest + shadow query): binary nodes 33.6 -> 22.9 node steps, two
// levels per step (what a 4-wide node is) 11.9; 9.7 -> 3.9 primitive phases; node visits x 1.06 (a child is sometimes
// tested with a bound that a sibling's hit would have shrunk), primitive tests x 1.3.  Measured instruction counts and
// frame times: EXPERIMENTS.md R4-1.
//
// Same answers as walk_ordered, whatever the order the items are processed in:
//   * shadow queries: the bound is fixed; a ray is occluded iff some reachable primitive accepts — a boolean OR.  An
//     accepted hit sets the ray's bound to -1, which makes its remaining items fail their tests;
//   * closest queries: the culling bound of a ray is (its nearest accepted distance) + (the tie radius), kept with an
//     LDS atomic minimum, and only ever shrinks — so no primitive whose hit lies within the tie radius of the FINAL
//     nearest distance is ever culled, and every such hit is appended to the ray's candidate list.  When the lists are
//     dry the ray's owner decides among its candidates exactly as test_slot decides a pair: candidates within the tie
//     radius of the nearest one are replayed in the reference's visiting order (rank): the first is accepted, each later
//     one only if its own leaf box still passes with the current distance as the bound and its distance is not larger
//     (triangle.cpp:82), the last accepted one is the hit.  The candidate SET does not depend on the processing order
//     (stale candidates — accepted before a much nearer hit was known — lie outside the tie radius and are ignored), so
//     neither does the result; the winner's record is then re-evaluated on the owner's own ray registers
//     (the same values test_slot would have copied: hit_from_record's argument).
//   * a ray that accepts more candidates than its list holds (pool_cands(); seen: up to 5 on cornell, 0.17 % of the rays
//     beyond 4) is walked again by its owner alone, the per-lane way (walk_ordered on a private stack).
// Sliver triangles: kSlivers below.  Not for scenes with opacity masks (their test draws random numbers DURING a walk, so
// the visiting order is part of the image: the reference-order walk, traversal.h).
//
// LDS per wavefront (kPoolWaveWords): 64 ray records of 12 words (16 with quadrics: the direction) — origin + bound |
// reciprocal direction + byte offsets of the near planes | shear + axis permutation —, 64 candidate counters, 64 x
// kPoolCands (distance, slot) pairs, 448 + 320 item slots of 16 bits (ray << 10 | node or slot; 32 bits, ray << 26, outside LDS).  The item counts
// live in scalar registers: the lists belong to ONE wavefront, no atomics on them.
#ifndef MCPT_POOL_WALK_H
#define MCPT_POOL_WALK_H

#include <type_traits>

#include "traversal.h"

namespace mcpt
{

// candidate hits a closest ray can hold (a ray with more walks alone — below).  6; 5 with merged queries, whose 13.3 KB per wavefront
// must leave room for three workgroups per CU (a ray through a vertex of valence 6 has six: dense smooth meshes — matpreview lost
// 2-3 % to the shorter list, so the unit without merged queries keeps six)
MCPT_HD constexpr uint32_t pool_cands(bool dual) { return dual ? 5u : 6u; }
constexpr uint32_t kPoolNodeItems = 448; // (ray, node) item slots: 384 in normal operation + 64 of head room (see below)
constexpr uint32_t kPoolNodeFull = 384;
constexpr uint32_t kPoolMaxDepth = 20;   // of the 4-wide hierarchy: 3 x depth <= the head room
#ifndef MCPT_POOL_PRIM_ITEMS
#define MCPT_POOL_PRIM_ITEMS 320
#endif
constexpr uint32_t kPoolPrimItems = MCPT_POOL_PRIM_ITEMS; // (ray, slot) item slots: < kPoolPrimAt waiting + 4 x 64 pushed by one node step
#ifndef MCPT_POOL_PRIM_AT
#define MCPT_POOL_PRIM_AT 64
#endif
constexpr uint32_t kPoolPrimAt = MCPT_POOL_PRIM_AT; // a primitive phase runs when this many slots wait (32 ... 128 measured on cornell: R4-9)
static_assert(kPoolPrimItems >= kPoolPrimAt + 256u, "a node step of 64 lanes can push 256 slots on top of the waiting ones");
constexpr uint32_t kPoolMaxRef = 1023;   // node and slot indices must fit 10 bits (16-bit items: scenes in LDS)
constexpr uint32_t kPoolMaxRefBig = (1u << 26) - 1u; // ... 26 bits (32-bit items: kFeatPoolBig)
constexpr uint32_t kPoolMaxRefDual = 511, kPoolMaxRefBigDual = (1u << 25) - 1u; // ... 9 / 25 bits with two rays per lane (merged queries)

// Scenes outside LDS (kBig): the node step reads the QUANTISED 4-wide form of the hierarchy (DeviceScene::wide_nodes: 64 bytes per
// node — four children's boxes as 8-bit offsets on the node's own grid, decoded boxes contain the exact ones, siblings adjacent, two
// nodes per 128-byte cache line) instead of the exact one (pool_nodes: 128 bytes), and the primitive step applies the primitive's
// own leaf-box test, which is what the exact hierarchy's last box test was (test_slot's kLeafCheck; commit.cpp, BuildWideNodes).
// Same reachability, same candidates, same answers — half the bytes per node visit through L2 / the memory-side cache, which is what
// the mesh kernels wait for (round 5).  -DMCPT_POOL_QUANT=0 builds the exact form for A/B measurements.
#ifndef MCPT_POOL_QUANT
#define MCPT_POOL_QUANT 1
#endif
// PACKED SLAB TESTS (round 6; measured, OFF).  A node step tests the boxes of kPer = 4 / 2 children against one ray: per child and
// axis the same `(plane - origin) * reciprocal` with the ray's two scalars.  gfx950 has packed FP32 instructions (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32: two IEEE single-precision operations per lane and instruction, each rounded like its scalar twin —
// tools/microbench/gen_pk_f32_modifiers.py checks every operand-select / negate combination against the scalar instructions);
// written on pairs of children the 48 subtractions and products of a four-child step are 24 instructions, the frames stay bit for
// bit — and the kernels get SLOWER: cornell 37.5 -> 39.1 ms, matpreview 413 -> 444 / 672 -> 716 ms
// (profiles/r06_ab_packed_slab_and_aligned_planes.jsonl): a packed instruction costs a wave64 the issue time of its two scalar
// halves, and the halves' results need a canonicalising v_max each before the max3 / min3 chains.  -DMCPT_POOL_PACKED=1 builds it.
// (What did pay in the same experiment: the plane sets of the exact records read as ONE aligned ds_read_b128 each — `Planes` below —
//  instead of two ds_read2_b32: cornell 38.9 -> 37.5 ms.)
#ifndef MCPT_POOL_PACKED
#define MCPT_POOL_PACKED 0
#endif
typedef float pool_float2 __attribute__((ext_vector_type(2)));

// NODE RECORDS IN LDS: 144 bytes apart (round 6).  A node step reads seven 16-byte plane sets of its item's node with ds_read_b128,
// whose bank is (address / 4) mod 64: records 128 bytes = 32 banks apart put every even node's planes on one set of banks and every
// odd node's on the other, so the 16 lanes of a read group, on ~8 different nodes, queued 4 deep behind each other (cornell:
// SQ_LDS_BANK_CONFLICT = 0.47 of the LDS's active cycles).  36 banks apart, nodes n and n + 16 are the first to share banks.
// 8: the records as they are stored in memory (A/B builds).
#ifndef MCPT_POOL_LDS_NODE_VECS
#define MCPT_POOL_LDS_NODE_VECS 9
#endif
constexpr uint32_t kPoolLdsNodeVecs = MCPT_POOL_LDS_NODE_VECS; // 16-byte vectors from one staged node record to the next (8 hold it)
static_assert(kPoolLdsNodeVecs >= 8, "a node record is 8 vectors");

#ifndef MCPT_POOL_CHILD_PARALLEL
#define MCPT_POOL_CHILD_PARALLEL 1 // node steps with 4 / 2 lanes per item when the items are few (walk_pool); 0: always one lane per item
#endif

// The leaf-box test of a TRIANGLE slot on the slot's own copy of the vertices (reference_leaf_box_passes reads tri_pos: the same
// values, one more gather): triangle.cpp:9-15's box = the bounds of the three vertices, aabb.cpp:29-48's slab test with `t_max`.
MCPT_HD bool slot_leaf_box_passes(const float4 *p, Ray ray, float t_max)
{
    ray.t_max = t_max;
    const V3 lo = vmin(vmin(xyz(p[0]), xyz(p[1])), xyz(p[2])), hi = vmax(vmax(xyz(p[0]), xyz(p[1])), xyz(p[2]));
    return box_hit(float4{lo.x, lo.y, lo.z, 0.0f}, float4{hi.x, hi.y, hi.z, 0.0f}, ray);
}

MCPT_HD constexpr uint32_t pool_ray_words(bool analytic) { return analytic ? 16u : 12u; }
// MERGED QUERIES (round 5, path_core.h path_step_merged): a lane brings up to TWO rays to one query — the closest-hit query of its
// path's next segment and the shadow query its previous vertex left pending (nothing between the two draws a random number, so they
// can be walked together: half the query rounds per path, twice the items per step).  `dual`: 128 ray records and counters per
// wavefront (record 64 + lane = the lane's shadow ray), one reference bit less in an item, a longer node list.
constexpr uint32_t kPoolNodeItemsDual = 640, kPoolNodeFullDual = 576;
MCPT_HD constexpr uint32_t pool_wave_words(bool analytic, bool big = false, bool dual = false)
{
    return (dual ? 128u : 64u) * pool_ray_words(analytic) + (dual ? 128u : 64u) + 64u * pool_cands(dual) * 2u +
           ((dual ? kPoolNodeItemsDual : kPoolNodeItems) + kPoolPrimItems) / (big ? 1u : 2u);
}

// The probe of one primitive slot WITHOUT any bound: does the ray's line hit it at t >= kEpsDistance, and where
// (test_slot's first half).
template <bool kAnalytic>
MCPT_HD SlotHit probe_slot(const DeviceScene &sc, const float4 *p, const Ray &ray)
{
    SlotHit h;
    const uint32_t inst = as_uint(p[1].w);
    if (!kAnalytic || sc.instances[inst].kind == kInstTriangles)
        return triangle_probe(p, ray);
    const InstanceRec &rec = sc.instances[inst];
    const uint32_t prim = as_uint(p[0].w);
    Ray probe = ray;
    probe.t_max = kMaxFloat;
    HitRaw cand;
    cand.a = cand.b = cand.c = 0.0f, cand.inside = false;
    uint32_t unused_rng = 0;
    if (rec.kind == kInstSphere)
        h.hit = sphere_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
    else if (rec.kind == kInstDisk)
        h.hit = disk_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
    else
        h.hit = cylinder_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
    h.t = probe.t_max, h.a = cand.a, h.b = cand.b, h.c = cand.c, h.inside = cand.inside;
    return h;
}

#if MCPT_WAVE_CODE

__device__ __forceinline__ void pool_sync()
{
    // the lists and records are written and read by the lanes of ONE wavefront: LDS executes a wavefront's accesses in
    // order, what is needed is that the compiler keeps them in order
#if defined(MCPT_POOL_SYNC_STRONG) // (A/B builds: every outstanding LDS access retired before the next one is issued)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
#else
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#endif
    __builtin_amdgcn_wave_barrier();
}

// base + the number of lanes below this one that are set in `mask` (v_mbcnt adds its third operand)
__device__ __forceinline__ uint32_t pool_rank(unsigned long long mask, uint32_t base = 0u)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), base));
}

// `pool`: the calling wavefront's kPoolWaveWords words of LDS.  Every lane that calls is a WORKER; the ones with
// `has_ray` also bring a ray — a lane without a path of its own (its pixel is finished, its sample ended at this vertex,
// the launch gave it none: RenderJob::lane_spread) helps the others' rays along, which is what shortens a pixel's chain
// when lanes are idle.  Returns whether the lane's ray hit anything; closest queries: `hit` and ray.t_max describe it.
// kSlivers (scenes with sliver triangles, IntegratorRec::walk_sliver_reach > 0; test_slot's rules): a sliver's leaf box in
// the hierarchy is a grown one, so its hit counts only if the reference's own leaf box lets the ray in; the radius within
// which two hits are candidates of one another is the sliver reach when one of them is a sliver and the tie radius
// otherwise, and an accepted hit shrinks the bound to its distance + ITS radius (a candidate carries its sliver flag in
// bit 31 of its slot word).  (One radius for everything — the reach — was tried first: in dense geometry, dragon/scene.xml's
// cloth, a dozen ordinary hits lie within 0.06 of each other and the candidate lists overflowed.)
// kDual (merged queries): the lane may bring a SECOND ray, a shadow ray (`shadow`, `has_shadow`; kAny must be false: the first ray is
// a closest query); `*occluded` tells whether that one hit anything.
template <bool kAny, bool kAnalytic, bool kCount, bool kBig = false, bool kSlivers = false, bool kDual = false>
__device__ __forceinline__ bool walk_pool(const DeviceScene &sc, uint32_t *pool, bool has_ray, Ray &ray, HitRaw &hit, TraceStats &stats, bool has_shadow = false,
                                          const Ray *shadow = nullptr, bool *occluded = nullptr)
{
    static_assert(!(kDual && kAny), "merged queries: the first ray is the closest query");
    using Item = typename std::conditional<kBig, uint32_t, uint16_t>::type; // ray << kRefBits | node or slot
    constexpr uint32_t kRefBits = (kBig ? 26u : 10u) - (kDual ? 1u : 0u), kRefMask = (1u << kRefBits) - 1u;
    constexpr uint32_t kPoolCands = pool_cands(kDual);
    constexpr uint32_t kRecords = kDual ? 128u : 64u, kNodeItems = kDual ? kPoolNodeItemsDual : kPoolNodeItems, kNodeFull = kDual ? kPoolNodeFullDual : kPoolNodeFull;
    constexpr bool kQuant = kBig && (MCPT_POOL_QUANT != 0); // the quantised node records + the leaf-box test at the primitive
    constexpr bool kPairs = kQuant && (MCPT_POOL_PAIRS != 0); // (experiment builds) node items name pairs of records: device_scene.h
    if (sc.integrator.n_walk_nodes == 0)
        return false;
    constexpr uint32_t kRayVecs = pool_ray_words(kAnalytic) / 4u;
    // (lockstep host build / experiment builds: the area holds nothing a query may rely on — wave_target.h)
    MCPT_POOL_POISON(pool, pool_wave_words(kAnalytic, kBig, kDual));
    const uint32_t lane = __lane_id();
    const unsigned long long workers = __ballot(1);
    const uint32_t rank = pool_rank(workers), n_workers = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__popcll(workers))));
    if (kPairs && n_workers < 2u)
        __builtin_trap(); // (a pair of records takes two lanes: every caller brings whole wavefronts)
    float4 *rays = reinterpret_cast<float4 *>(pool);
    uint32_t *counts = pool + kRecords * pool_ray_words(kAnalytic);
    uint2 *cands = reinterpret_cast<uint2 *>(counts + kRecords);
    Item *node_items = reinterpret_cast<Item *>(counts + kRecords + 64u * kPoolCands * 2u), *prim_items = node_items + kNodeItems;
    const float tie = sc.integrator.walk_tie, reach = kSlivers ? sc.integrator.walk_sliver_reach : sc.integrator.walk_tie;

    // ---- the lane's ray becomes a record ----
    const unsigned long long m_rays = __ballot(has_ray), m_shadows = kDual ? __ballot(has_shadow) : 0ull;
    if (kDual && occluded)
        *occluded = false;
    if ((m_rays | m_shadows) == 0)
        return false;
    auto write_record = [&](uint32_t at, const Ray &q)
    {
        const uint32_t nx = q.dir_rcp.x > 0 ? 0u : 48u, ny = q.dir_rcp.y > 0 ? 16u : 64u, nz = q.dir_rcp.z > 0 ? 32u : 80u;
        const uint32_t pack = nx | (ny << 8) | (nz << 16);
        const uint32_t axes = static_cast<uint32_t>(q.kx) | (static_cast<uint32_t>(q.ky) << 2) | (static_cast<uint32_t>(q.kz) << 4);
        rays[kRayVecs * at + 0] = float4{q.origin.x, q.origin.y, q.origin.z, q.t_max};
        rays[kRayVecs * at + 1] = float4{q.dir_rcp.x, q.dir_rcp.y, q.dir_rcp.z, __uint_as_float(pack)};
        rays[kRayVecs * at + 2] = float4{q.shear.x, q.shear.y, q.shear.z, __uint_as_float(axes)};
        if (kAnalytic)
            rays[kRayVecs * at + 3] = float4{q.dir.x, q.dir.y, q.dir.z, 0.0f};
        counts[at] = 0;
    };
    if (kDual && has_shadow)
    {
        // (record 64 + lane; its top-node item behind the closest rays')
        write_record(64u + lane, *shadow);
        node_items[static_cast<uint32_t>(__popcll(m_rays)) + pool_rank(m_shadows)] = static_cast<Item>((64u + lane) << kRefBits);
    }
    if (has_ray && kDual)
    {
        write_record(lane, ray);
        node_items[pool_rank(m_rays)] = static_cast<Item>(lane << kRefBits);
    }
    if (has_ray && !kDual)
    {
        // byte offsets, inside a node's 128-byte record {lo.x lo.y lo.z hi.x hi.y hi.z of the four children | references}, of
        // the planes the ray enters through (walk_ordered's sign-addressed reads): x: 0 or 48, y: 16 or 64, z: 32 or 80
        const uint32_t nx = ray.dir_rcp.x > 0 ? 0u : 48u, ny = ray.dir_rcp.y > 0 ? 16u : 64u, nz = ray.dir_rcp.z > 0 ? 32u : 80u;
        const uint32_t pack = nx | (ny << 8) | (nz << 16);
        const uint32_t axes = static_cast<uint32_t>(ray.kx) | (static_cast<uint32_t>(ray.ky) << 2) | (static_cast<uint32_t>(ray.kz) << 4);
        rays[kRayVecs * lane + 0] = float4{ray.origin.x, ray.origin.y, ray.origin.z, ray.t_max};
        rays[kRayVecs * lane + 1] = float4{ray.dir_rcp.x, ray.dir_rcp.y, ray.dir_rcp.z, __uint_as_float(pack)};
        rays[kRayVecs * lane + 2] = float4{ray.shear.x, ray.shear.y, ray.shear.z, __uint_as_float(axes)};
        if (kAnalytic)
            rays[kRayVecs * lane + 3] = float4{ray.dir.x, ray.dir.y, ray.dir.z, 0.0f};
        counts[lane] = 0;
        node_items[pool_rank(m_rays)] = static_cast<Item>(lane << kRefBits); // (ray, top node)
    }
    // (wavefront-uniform values, kept in scalar registers: `uni` tells the compiler so where it cannot see it)
    auto uni = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
    uint32_t n_nodes = uni(static_cast<uint32_t>(__popcll(m_rays) + __popcll(m_shadows))), n_prims = 0;
    // One node step over the top `k` items with G lanes per item (G = 1, 2, 4), each lane testing kPer = 4 / G children of its
    // item's node.  Everything inside ONE region of the working lanes — ballots included, they only see those lanes: predicates
    // that leave the region would travel as 0 / 1 words through vector registers.
    auto node_step = [&](auto g_tag, const uint32_t k)
    {
        // (kPairs, experiment builds: an item names a PAIR of records, G = 2 / 4 / 8 lanes per item, G / 2 of them on each record)
        constexpr uint32_t G = decltype(g_tag)::value, Gh = kPairs ? G / 2u : G, kPer = 4u / Gh;
        static_assert(!kPairs || G >= 2u, "a pair of records takes at least two lanes");
        uint32_t next_nodes = 0, next_prims = 0;
        if (rank < k * G)
        {
            MCPT_WAVE_REGION();
            const uint32_t half = kPairs ? (rank & (G - 1u)) / Gh : 0u;
            const uint32_t sub = rank & (Gh - 1u), first = sub * kPer; // this lane's children: first ... first + kPer - 1 of its record (Gh = 1: all four)
            if (kCount)
                stats.node_tests += kPer;
            const uint32_t item = node_items[n_nodes - 1u - rank / G];
            const uint32_t ray_bits = item & ~kRefMask, r = item >> kRefBits, node = item & kRefMask;
            const float4 a = rays[kRayVecs * r], b = rays[kRayVecs * r + 1];
            const uint32_t pack = __float_as_uint(b.w);
            float enter[kPer], leave[kPer];
            uint32_t ref[kPer];
            if constexpr (kQuant)
            {
                // one 64-byte record (16-byte reads of one half cache line): grid origin + exponents | references | the planes
                // lo.x lo.y lo.z hi.x | hi.y hi.z, one byte per child.  Planes are decoded to world space — origin + 2^e * q, the
                // operations the quantiser verified (the product is exact, so the fused form rounds once, to the same value) — and
                // go through the same slab test as the exact boxes they contain.
                const uint4 *wq = sc.wide_nodes + (kPairs ? 8u : 4u) * static_cast<size_t>(node) + 4u * half;
                const uint4 n0 = wq[0], n2 = wq[2], n3 = wq[3];
                if constexpr (Gh == 1)
                {
                    const uint4 refs = wq[1];
                    ref[0] = refs.x, ref[1] = refs.y, ref[2] = refs.z, ref[3] = refs.w;
                }
                else if constexpr (Gh == 2)
                {
                    const uint2 refs = reinterpret_cast<const uint2 *>(wq + 1)[sub];
                    ref[0] = refs.x, ref[1] = refs.y;
                }
                else
                    ref[0] = reinterpret_cast<const uint32_t *>(wq + 1)[sub];
                const float gx = __uint_as_float(n0.x), gy = __uint_as_float(n0.y), gz = __uint_as_float(n0.z);
                const float sx = __uint_as_float((n0.w & 0xFFu) << 23), sy = __uint_as_float(((n0.w >> 8) & 0xFFu) << 23), sz = __uint_as_float(((n0.w >> 16) & 0xFFu) << 23);
                const bool px = (pack & 0xffu) == 0u, py = ((pack >> 8) & 0xffu) == 16u, pz = ((pack >> 16) & 0xffu) == 32u; // (dir_rcp > 0)
                const uint32_t near_x = px ? n2.x : n2.w, far_x = px ? n2.w : n2.x;
                const uint32_t near_y = py ? n2.y : n3.x, far_y = py ? n3.x : n2.y;
                const uint32_t near_z = pz ? n2.z : n3.y, far_z = pz ? n3.y : n2.z;
                if constexpr (MCPT_POOL_PACKED != 0 && kPer >= 2)
                {
                    // (pairs of children through the packed FP32 instructions: decode — one fused multiply-add per plane, as the
                    //  quantiser verified it — subtraction and product for two children at a time)
#pragma unroll
                    for (uint32_t j = 0; j < kPer; j += 2)
                    {
                        const uint32_t shift = 8u * (first + j);
                        auto slab = [shift](uint32_t word, float scale, float grid, float origin, float rcp)
                        {
                            const pool_float2 q = {static_cast<float>((word >> shift) & 0xFFu), static_cast<float>((word >> (shift + 8u)) & 0xFFu)};
                            const pool_float2 plane = __builtin_elementwise_fma(pool_float2{scale, scale}, q, pool_float2{grid, grid});
                            return (plane - pool_float2{origin, origin}) * pool_float2{rcp, rcp};
                        };
                        const pool_float2 ex = slab(near_x, sx, gx, a.x, b.x), ey = slab(near_y, sy, gy, a.y, b.y), ez = slab(near_z, sz, gz, a.z, b.z);
                        const pool_float2 lx = slab(far_x, sx, gx, a.x, b.x), ly = slab(far_y, sy, gy, a.y, b.y), lz = slab(far_z, sz, gz, a.z, b.z);
                        enter[j] = fmaxf(fmaxf(fmaxf(kEpsDistance, ex.x), ey.x), ez.x), enter[j + 1] = fmaxf(fmaxf(fmaxf(kEpsDistance, ex.y), ey.y), ez.y);
                        leave[j] = fminf(fminf(fminf(a.w, lx.x), ly.x), lz.x), leave[j + 1] = fminf(fminf(fminf(a.w, lx.y), ly.y), lz.y);
                    }
                }
                else
                {
#pragma unroll
                    for (uint32_t j = 0; j < kPer; ++j)
                    {
                        const uint32_t shift = 8u * (first + j); // (a constant when G = 1)
                        auto plane = [shift](uint32_t word, float scale, float origin) { return __builtin_fmaf(scale, static_cast<float>((word >> shift) & 0xFFu), origin); };
                        enter[j] = fmaxf(fmaxf(fmaxf(kEpsDistance, (plane(near_x, sx, gx) - a.x) * b.x), (plane(near_y, sy, gy) - a.y) * b.y), (plane(near_z, sz, gz) - a.z) * b.z);
                        leave[j] = fminf(fminf(fminf(a.w, (plane(far_x, sx, gx) - a.x) * b.x), (plane(far_y, sy, gy) - a.y) * b.y), (plane(far_z, sz, gz) - a.z) * b.z);
                    }
                }
            }
            else
            {
                // the planes the ray enters / leaves through, of this lane's children: lo.x at 0, lo.y 16, lo.z 32, hi.x 48, hi.y 64,
                // hi.z 80 of the 128-byte record, four children each; the references at 96
                // (the exact form outside LDS exists in A/B builds only, MCPT_POOL_QUANT=0: records as stored, 128 bytes apart)
                const char *w = reinterpret_cast<const char *>(sc.pool_nodes) + (kBig ? 128u : 16u * kPoolLdsNodeVecs) * node + 4u * first;
                const uint32_t ox = pack & 0xffu, oy = (pack >> 8) & 0xffu, oz = (pack >> 16) & 0xffu;
                struct alignas(4u * kPer) Planes // (the planes of this lane's children: one 16- / 8- / 4-byte LDS read each)
                {
                    float v[kPer];
                };
                // (every offset is a multiple of 4 kPer bytes inside a 128-byte record: one ds_read_b128 / _b64 per plane set)
                auto planes_at = [&](uint32_t offset) { return *reinterpret_cast<const Planes *>(w + offset); };
                const Planes nx = planes_at(ox), fx = planes_at(48u - ox);
                const Planes ny = planes_at(oy), fy = planes_at(80u - oy);
                const Planes nz = planes_at(oz), fz = planes_at(112u - oz);
                const Planes rf = planes_at(96u);
                if constexpr (MCPT_POOL_PACKED != 0 && kPer >= 2)
                {
#pragma unroll
                    for (uint32_t j = 0; j < kPer; j += 2)
                    {
                        auto slab = [](const Planes &p, uint32_t at, float origin, float rcp)
                        { return (pool_float2{p.v[at], p.v[at + 1]} - pool_float2{origin, origin}) * pool_float2{rcp, rcp}; };
                        const pool_float2 ex = slab(nx, j, a.x, b.x), ey = slab(ny, j, a.y, b.y), ez = slab(nz, j, a.z, b.z);
                        const pool_float2 lx = slab(fx, j, a.x, b.x), ly = slab(fy, j, a.y, b.y), lz = slab(fz, j, a.z, b.z);
                        enter[j] = fmaxf(fmaxf(fmaxf(kEpsDistance, ex.x), ey.x), ez.x), enter[j + 1] = fmaxf(fmaxf(fmaxf(kEpsDistance, ex.y), ey.y), ez.y);
                        leave[j] = fminf(fminf(fminf(a.w, lx.x), ly.x), lz.x), leave[j + 1] = fminf(fminf(fminf(a.w, lx.y), ly.y), lz.y);
                    }
                }
                else
                {
#pragma unroll
                    for (uint32_t j = 0; j < kPer; ++j)
                    {
                        enter[j] = fmaxf(fmaxf(fmaxf(kEpsDistance, (nx.v[j] - a.x) * b.x), (ny.v[j] - a.y) * b.y), (nz.v[j] - a.z) * b.z);
                        leave[j] = fminf(fminf(fminf(a.w, (fx.v[j] - a.x) * b.x), (fy.v[j] - a.y) * b.y), (fz.v[j] - a.z) * b.z);
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < kPer; ++j)
                    ref[j] = __float_as_uint(rf.v[j]);
            }
            uint32_t at_nodes = n_nodes - k, at_prims = n_prims;
#pragma unroll
            for (uint32_t j = 0; j < kPer; ++j)
            {
                // (quantised form: an unused child — kWalkDone, an inverted box — is never entered, whatever a degenerate grid or a NaN
                //  ray says; exact form: an unused child names slot 0 behind a box that only a NaN ray "enters", like the reference's)
                const bool hit_c = enter[j] <= leave[j] && (!kQuant || ref[j] != kWalkDone), leaf_c = static_cast<int32_t>(ref[j]) < 0; // kWalkLeaf = the sign bit
                // (ballots of plain comparisons, combined as 64-bit masks: a ballot of a combined predicate goes through a
                //  vector register as a 0 / 1 word)
                const unsigned long long b_hit = __ballot(hit_c), b_leaf = __ballot(leaf_c);
                const unsigned long long m_node = b_hit & ~b_leaf, m_leaf = b_hit & b_leaf;
                if (hit_c && !leaf_c)
                    (node_items + at_nodes)[pool_rank(m_node)] = static_cast<Item>(ray_bits | ref[j]);
                if (hit_c && leaf_c)
                    (prim_items + at_prims)[pool_rank(m_leaf)] = static_cast<Item>(ray_bits | (ref[j] & kRefMask));
                at_nodes += static_cast<uint32_t>(__popcll(m_node)), at_prims += static_cast<uint32_t>(__popcll(m_leaf));
            }
            next_nodes = at_nodes, next_prims = at_prims;
        }
        n_nodes = uni(next_nodes), n_prims = uni(next_prims); // (the first active lane has rank 0 < k G: it took part)
    };
    {
        pool_sync();
        for (;;)
        {
            if (n_prims >= kPoolPrimAt || n_nodes == 0)
            {
                if (n_prims == 0)
                    break;
                // ---- primitive phase: the top min(workers, waiting) slots, one per lane ----
                const uint32_t k = uni(n_prims < n_workers ? n_prims : n_workers);
                if (kCount && rank == 0)
                    ++stats.wave_prim_steps;
                if (rank < k)
                {
                    if (kCount)
                        ++stats.prim_tests;
                    const uint32_t item = prim_items[n_prims - 1u - rank], r = item >> kRefBits, slot = item & kRefMask;
                    const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(slot);
                    float4 a;   // origin + bound
                    Ray q;      // (the whole ray: quadrics and the sliver rules need it)
                    SlotHit h;
                    if constexpr (!kAnalytic && !kSlivers)
                    {
                        // triangles only: the vertices' and the origin's components are READ in the ray's axis order (word kx, ky, kz of
                        // each record) instead of being selected from whole records — 18 selects per test become addresses
                        const float4 c = rays[kRayVecs * r + 2];
                        const uint32_t axes = __float_as_uint(c.w), kx = axes & 3u, ky = (axes >> 2) & 3u, kz = (axes >> 4) & 3u;
                        const float *ro = reinterpret_cast<const float *>(&rays[kRayVecs * r]), *pf = reinterpret_cast<const float *>(p);
                        const float ox = ro[kx], oy = ro[ky], oz = ro[kz];
                        a.x = a.y = a.z = 0.0f, a.w = ro[3];
                        const float Ax = pf[kx] - ox, Ay = pf[ky] - oy, Az = pf[kz] - oz, Bx = pf[4 + kx] - ox, By = pf[4 + ky] - oy, Bz = pf[4 + kz] - oz;
                        const float Cx = pf[8 + kx] - ox, Cy = pf[8 + ky] - oy, Cz = pf[8 + kz] - oz;
                        h = triangle_probe_permuted(Ax, Ay, Az, Bx, By, Bz, Cx, Cy, Cz, V3{c.x, c.y, c.z});
                        if (kQuant)
                        {
                            // the primitive's own leaf box (box_hit on the bounds of the three vertices) under the ray's current
                            // bound, on the offsets that are already here: min / max commute with subtracting the origin (rounding is
                            // monotone) and the slab test treats the axes alike, so their permuted order changes nothing
                            const float *rb = reinterpret_cast<const float *>(&rays[kRayVecs * r + 1]);
                            const float ix = rb[kx], iy = rb[ky], iz = rb[kz];
                            const float lx = fminf(fminf(Ax, Bx), Cx), ly = fminf(fminf(Ay, By), Cy), lz = fminf(fminf(Az, Bz), Cz);
                            const float hx = fmaxf(fmaxf(Ax, Bx), Cx), hy = fmaxf(fmaxf(Ay, By), Cy), hz = fmaxf(fmaxf(Az, Bz), Cz);
                            const bool px = ix > 0, py = iy > 0, pz = iz > 0;
                            const float t_enter = fmaxf(fmaxf(fmaxf(kEpsDistance, (px ? lx : hx) * ix), (py ? ly : hy) * iy), (pz ? lz : hz) * iz);
                            const float t_exit = fminf(fminf(fminf(a.w, (px ? hx : lx) * ix), (py ? hy : ly) * iy), (pz ? hz : lz) * iz);
                            h.hit = h.hit && t_enter <= t_exit;
                        }
                    }
                    else
                    {
                        a = rays[kRayVecs * r];
                        const float4 b = rays[kRayVecs * r + 1], c = rays[kRayVecs * r + 2];
                        q.origin = V3{a.x, a.y, a.z}, q.dir_rcp = V3{b.x, b.y, b.z}, q.shear = V3{c.x, c.y, c.z}, q.t_max = a.w;
                        const uint32_t axes = __float_as_uint(c.w);
                        q.kx = static_cast<int>(axes & 3u), q.ky = static_cast<int>((axes >> 2) & 3u), q.kz = static_cast<int>((axes >> 4) & 3u);
                        q.dir = V3{0.0f, 0.0f, 0.0f};
                        if (kAnalytic)
                        {
                            const float4 d = rays[kRayVecs * r + 3];
                            q.dir = V3{d.x, d.y, d.z};
                        }
                        h = probe_slot<kAnalytic>(sc, p, q);
                        // (a sliver's box in the hierarchy is a grown one in both forms and its own rules below decide)
                        if (kQuant && h.hit && !(kSlivers && (as_uint(p[2].w) & kWalkSliver) != 0))
                            h.hit = (!kAnalytic || sc.instances[as_uint(p[1].w)].kind == kInstTriangles) ? slot_leaf_box_passes(p, q, a.w)
                                                                                                        : reference_leaf_box_passes<kAnalytic>(sc, as_uint(p[1].w), as_uint(p[0].w), q, a.w);
                    }
                    float *bound = reinterpret_cast<float *>(&rays[kRayVecs * r]) + 3;
                    // (a sliver reached through its grown box: would the reference's own leaf box let the ray in?)
                    const bool sliver = kSlivers && (as_uint(p[2].w) & kWalkSliver) != 0;
                    const bool any = kDual ? r >= 64u : kAny; // (merged queries: the upper records are the shadow rays)
                    if (any)
                    {
                        if (h.hit && !(h.t > a.w) && (!sliver || reference_leaf_box_passes<kAnalytic>(sc, as_uint(p[1].w), as_uint(p[0].w), q, a.w))) // (test_slot, kAny)
                        {
                            *bound = -1.0f; // occluded: whatever of the ray is still listed fails its test
                            counts[r] = 1u;
                        }
                    }
                    else if (h.hit && h.t <= a.w && (!sliver || reference_leaf_box_passes<kAnalytic>(sc, as_uint(p[1].w), as_uint(p[0].w), q, kMaxFloat)))
                    {
                        // the bound shrinks to this distance + the hit's radius (monotone: positive floats order like their bits)
                        const uint32_t before = atomicMin(reinterpret_cast<uint32_t *>(bound), __float_as_uint(h.t + (sliver ? reach : tie)));
                        if (!(h.t > __uint_as_float(before)))
                        {
                            const uint32_t at = atomicAdd(&counts[r], 1u);
                            if (at < kPoolCands)
                                cands[kPoolCands * r + at] = uint2{__float_as_uint(h.t), slot | (sliver ? kWalkSliver : 0u)};
                        }
                    }
                }
                n_prims = uni(n_prims - k);
                pool_sync();
                continue;
            }
            // ---- node phase: the top items of the node list; the FOUR children of a node (DeviceScene::pool_nodes / wide_nodes) tested,
            //      the hit ones pushed (in the order the node stores them: a model of this schedule with near children on top of far
            //      ones gave 11.9 instead of 12.4 steps per round on cornell — not worth a sort per item) ----
            // (head room: a step with k items grows the list by at most 3 k.  Below kPoolNodeFull every worker takes an item
            //  if that fits; above, ONE does — a depth-first walk, which adds at most 3 x the tree's depth to the list)
            // (pairs: two lanes per item at least, an item lists up to eight children — seven more than it took away)
            constexpr uint32_t kGrowth = kPairs ? 7u : 3u, kLanesAtLeast = kPairs ? 2u : 1u, kFull = kPairs ? kNodeItems - kGrowth * kPoolMaxDepth : kNodeFull;
            const uint32_t takers = uni(n_workers / kLanesAtLeast);
            uint32_t k = uni(n_nodes < takers ? n_nodes : takers);
            const uint32_t room = uni(n_nodes < kFull ? (kFull - n_nodes + kGrowth - 1u) / kGrowth : 1u);
            k = uni(k < room ? k : room);
            if (kCount && rank == 0)
                ++stats.wave_node_steps;
            // CHILD-PARALLEL steps (round 5).  With k items and 64 lanes, one lane per item leaves 64 - k lanes idle while the k busy ones
            // test four children one after the other.  When the items are few — the start and the tail of every query, and every step of
            // a wavefront that holds a handful of paths (a rank's small tile share, the end of a frame: the lanes without a path are
            // workers here) — G = 4 or 2 lanes share an item and test 4 / G children each: the same tests, pushes and items in a
            // quarter / half of the instructions, which is what a near-empty wavefront's step costs (its chain of dependent
            // instructions), and what a VALU-bound kernel pays per item.  Which lane tests which child changes the ORDER of the pushed
            // items only; the walk's answers do not depend on it.
            if constexpr (kPairs)
            {
                if (k * 8u <= n_workers)
                    node_step(std::integral_constant<uint32_t, 8u>{}, k);
                else if (k * 4u <= n_workers)
                    node_step(std::integral_constant<uint32_t, 4u>{}, k);
                else
                    node_step(std::integral_constant<uint32_t, 2u>{}, k);
            }
            else
            {
#if MCPT_POOL_CHILD_PARALLEL
                if (k * 4u <= n_workers)
                    node_step(std::integral_constant<uint32_t, 4u>{}, k);
                else if (k * 2u <= n_workers)
                    node_step(std::integral_constant<uint32_t, 2u>{}, k);
                else
#endif
                    node_step(std::integral_constant<uint32_t, 1u>{}, k);
            }
            pool_sync();
        }
    }
    if (kDual && occluded && has_shadow)
        *occluded = counts[64u + lane] != 0;
    if (!has_ray)
        return false;
    if (kAny)
        return counts[lane] != 0;

    // ---- closest query: the owner decides among its candidates ----
    uint32_t n = counts[lane];
    if (n == 0)
        return false;
    if (n > kPoolCands)
    {
        // more accepted hits than the list holds (many near-coincident surfaces: the apex of dragon/scene.xml's stand-in wings,
        // 20 sheets through one point): this lane walks its ray alone, the per-lane way, on a private stack
        MCPT_WAVE_REGION();
        uint32_t own_stack[kWalkStackMax];
        return walk_ordered<false, kAnalytic, false, kSlivers, 1u>(sc, own_stack, ray, hit, stats);
    }
    const uint2 *mine = cands + kPoolCands * lane;
    float t_min = __uint_as_float(mine[0].x);
    uint32_t winner = mine[0].y;
    for (uint32_t i = 1; i < n; ++i)
    {
        const float t = __uint_as_float(mine[i].x);
        if (t < t_min)
            t_min = t, winner = mine[i].y;
    }
    // the nearest hit's rivals: within the tie radius of it — the sliver reach if either of the two is a sliver
    const bool nearest_sliver = kSlivers && (winner & kWalkSliver) != 0;
    auto rival = [&](uint32_t i) { return fabsf(__uint_as_float(mine[i].x) - t_min) <= ((kSlivers && (nearest_sliver || (mine[i].y & kWalkSliver))) ? reach : tie); };
    uint32_t n_tied = 0;
    for (uint32_t i = 0; i < n; ++i)
        n_tied += rival(i) ? 1u : 0u;
    winner &= ~kWalkSliver;
    if (n_tied > 1)
    {
        // rare: several hits within the radius of the nearest — the reference's own sequence on them, in rank order
        // (test_slot's pair rule, applied along the whole list)
        uint32_t last_rank = 0;
        float cur_t = 0.0f;
        for (uint32_t step = 0; step < n_tied; ++step)
        {
            uint32_t pick = 0, pick_rank = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < n; ++i)
            {
                if (!rival(i))
                    continue;
                const uint32_t rk = as_uint(sc.walk_prims[3 * static_cast<size_t>(mine[i].y & ~kWalkSliver) + 2].w) & ~kWalkSliver;
                if ((step == 0 || rk > last_rank) && rk < pick_rank)
                    pick = i, pick_rank = rk;
            }
            const float t = __uint_as_float(mine[pick].x);
            const uint32_t slot = mine[pick].y & ~kWalkSliver;
            const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(slot);
            if (step == 0 || (!(t > cur_t) && reference_leaf_box_passes<kAnalytic>(sc, as_uint(p[1].w), as_uint(p[0].w), ray, cur_t)))
                cur_t = t, winner = slot;
            last_rank = pick_rank;
        }
    }
    const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(winner);
    const SlotHit h = probe_slot<kAnalytic>(sc, p, ray);
    hit.inst = as_uint(p[1].w), hit.prim = as_uint(p[0].w), hit.a = h.a, hit.b = h.b, hit.c = h.c, hit.inside = h.inside;
    ray.t_max = h.t;
    return true;
}

#endif // MCPT_WAVE_CODE

} // namespace mcpt

#endif // MCPT_POOL_WALK_H
