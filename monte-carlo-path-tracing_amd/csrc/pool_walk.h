// The wavefront-cooperative ray query of LDS-resident scenes (round 4).
//
// What it cures.  walk_ordered (traversal.h) gives a ray to a lane: a wavefront's query lasts as long as its slowest
// lane's walk (cornell: a ray needs ~10 node visits, the wavefront runs ~34 node steps and ~10 primitive phases per round:
// 29 % / 24 % of the lanes do useful work) — the reference's per-thread nested stack walk has the same shape
// (tlas.cpp:13-76, blas.cpp:18-77 inside the megakernel, renderer.cpp:88-95).  Here the rays of a wavefront's query
// (one per active lane) are RECORDS in LDS and the work is a shared LIFO of (ray, node) items and a second one of
// (ray, primitive slot) items: every step, every lane takes the next item, whichever ray it belongs to — a box test
// step pushes the children it hits (far children below near ones), a primitive step tests one slot.  All lanes stay
// busy until the lists run dry; a primitive phase runs when a wavefront's worth of slots waits (or nothing else is left),
// so both phases run full.  tests/emu's model of exactly this schedule (Pool2Model): 33.6 -> 22.9 node steps and
// 9.7 -> 3.9 primitive phases per round on cornell, node visits x 1.05 (a far child is sometimes tested with a bound that
// the near child's hit would have shrunk), primitive tests x 1.28.
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
//   * a ray that accepts more candidates than its list holds (kPoolCands; seen: up to 5 on cornell, 0.17 % of the rays
//     beyond 4) is walked a second time with its final bound as the initial one: then only the hits within the tie
//     radius of the nearest are accepted at all.
// Not for scenes with slivers (kFeatSlivers: their reachability rules are test_slot's) or opacity masks.
//
// LDS per wavefront (kPoolWaveWords): 64 ray records of 12 words (16 with quadrics: the direction) — origin + bound |
// reciprocal direction + byte offsets of the near planes | shear + axis permutation —, 64 candidate counters, 64 x
// kPoolCands (distance, slot) pairs, 448 + 192 item slots of 16 bits (ray << 10 | node or slot).  The item counts
// live in scalar registers: the lists belong to ONE wavefront, no atomics on them.
#ifndef MCPT_POOL_WALK_H
#define MCPT_POOL_WALK_H

#include "traversal.h"

namespace mcpt
{

constexpr uint32_t kPoolCands = 6;       // candidate hits a closest ray can hold
constexpr uint32_t kPoolNodeItems = 448; // (ray, node) item slots: 384 in normal operation + 64 of head room (see below)
constexpr uint32_t kPoolNodeFull = 384;
constexpr uint32_t kPoolPrimItems = 192; // (ray, slot) item slots: < kPoolPrimAt waiting + 2 x 64 pushed by one node step
constexpr uint32_t kPoolPrimAt = 64;     // a primitive phase runs when this many slots wait
constexpr uint32_t kPoolMaxRef = 1023;   // node and slot indices must fit 10 bits

MCPT_HD constexpr uint32_t pool_ray_words(bool analytic) { return analytic ? 16u : 12u; }
MCPT_HD constexpr uint32_t pool_wave_words(bool analytic)
{
    return 64u * pool_ray_words(analytic) + 64u + 64u * kPoolCands * 2u + (kPoolNodeItems + kPoolPrimItems) / 2u;
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

#if defined(__HIPCC__)

__device__ __forceinline__ void pool_sync()
{
    // the lists and records are written and read by the lanes of ONE wavefront: LDS executes a wavefront's accesses in
    // order, what is needed is that the compiler keeps them in order
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
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
template <bool kAny, bool kAnalytic, bool kCount>
__device__ __forceinline__ bool walk_pool(const DeviceScene &sc, uint32_t *pool, bool has_ray, Ray &ray, HitRaw &hit, TraceStats &stats)
{
    if (sc.integrator.n_walk_nodes == 0)
        return false;
    constexpr uint32_t kRayVecs = pool_ray_words(kAnalytic) / 4u;
    const uint32_t lane = __lane_id();
    const unsigned long long workers = __ballot(1);
    const uint32_t rank = pool_rank(workers), n_workers = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__popcll(workers))));
    float4 *rays = reinterpret_cast<float4 *>(pool);
    uint32_t *counts = pool + 64u * pool_ray_words(kAnalytic);
    uint2 *cands = reinterpret_cast<uint2 *>(counts + 64u);
    uint16_t *node_items = reinterpret_cast<uint16_t *>(counts + 64u + 64u * kPoolCands * 2u), *prim_items = node_items + kPoolNodeItems;
    const float tie = sc.integrator.walk_tie;

    // ---- the lane's ray becomes a record ----
    const unsigned long long m_rays = __ballot(has_ray);
    if (m_rays == 0)
        return false;
    if (has_ray)
    {
        // byte offsets, inside a node's 64-byte record {lo0 ref0 | hi0 ref1 | lo1 - | hi1 -}, of the planes the ray
        // enters through (walk_ordered's sign-addressed reads): x: 0 or 16, y: 4 or 20, z: 8 or 24
        const uint32_t nx = ray.dir_rcp.x > 0 ? 0u : 16u, ny = ray.dir_rcp.y > 0 ? 4u : 20u, nz = ray.dir_rcp.z > 0 ? 8u : 24u;
        const uint32_t pack = nx | (ny << 8) | (nz << 16);
        const uint32_t axes = static_cast<uint32_t>(ray.kx) | (static_cast<uint32_t>(ray.ky) << 2) | (static_cast<uint32_t>(ray.kz) << 4);
        rays[kRayVecs * lane + 0] = float4{ray.origin.x, ray.origin.y, ray.origin.z, ray.t_max};
        rays[kRayVecs * lane + 1] = float4{ray.dir_rcp.x, ray.dir_rcp.y, ray.dir_rcp.z, __uint_as_float(pack)};
        rays[kRayVecs * lane + 2] = float4{ray.shear.x, ray.shear.y, ray.shear.z, __uint_as_float(axes)};
        if (kAnalytic)
            rays[kRayVecs * lane + 3] = float4{ray.dir.x, ray.dir.y, ray.dir.z, 0.0f};
        counts[lane] = 0;
        node_items[pool_rank(m_rays)] = static_cast<uint16_t>(lane << 10); // (ray, top node)
    }
    // (wavefront-uniform values, kept in scalar registers: `uni` tells the compiler so where it cannot see it)
    auto uni = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
    uint32_t n_nodes = uni(static_cast<uint32_t>(__popcll(m_rays))), n_prims = 0;
    for (uint32_t pass = 0;; ++pass)
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
                    const uint32_t item = prim_items[n_prims - 1u - rank], r = item >> 10, slot = item & kPoolMaxRef;
                    const float4 a = rays[kRayVecs * r], b = rays[kRayVecs * r + 1], c = rays[kRayVecs * r + 2];
                    Ray q;
                    q.origin = V3{a.x, a.y, a.z}, q.dir_rcp = V3{b.x, b.y, b.z}, q.shear = V3{c.x, c.y, c.z}, q.t_max = a.w;
                    const uint32_t axes = __float_as_uint(c.w);
                    q.kx = static_cast<int>(axes & 3u), q.ky = static_cast<int>((axes >> 2) & 3u), q.kz = static_cast<int>((axes >> 4) & 3u);
                    q.dir = V3{0.0f, 0.0f, 0.0f};
                    if (kAnalytic)
                    {
                        const float4 d = rays[kRayVecs * r + 3];
                        q.dir = V3{d.x, d.y, d.z};
                    }
                    const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(slot);
                    const SlotHit h = probe_slot<kAnalytic>(sc, p, q);
                    float *bound = reinterpret_cast<float *>(&rays[kRayVecs * r]) + 3;
                    if (kAny)
                    {
                        if (h.hit && !(h.t > a.w)) // (test_slot, kAny)
                        {
                            *bound = -1.0f; // occluded: whatever of the ray is still listed fails its test
                            counts[r] = 1u;
                        }
                    }
                    else if (h.hit && h.t <= a.w)
                    {
                        // the bound shrinks to this distance + the tie radius (monotone: positive floats order like their bits)
                        const uint32_t before = atomicMin(reinterpret_cast<uint32_t *>(bound), __float_as_uint(h.t + tie));
                        if (!(h.t > __uint_as_float(before)))
                        {
                            const uint32_t at = atomicAdd(&counts[r], 1u);
                            if (at < kPoolCands)
                                cands[kPoolCands * r + at] = uint2{__float_as_uint(h.t), slot};
                        }
                    }
                }
                n_prims = uni(n_prims - k);
                pool_sync();
                continue;
            }
            // ---- node phase: the top items, one per lane; both children of a node tested, the hit ones pushed ----
            // (head room: a step with k lanes grows the list by at most k.  Below kPoolNodeFull every worker takes an item;
            //  above, fewer do, down to ONE — a depth-first walk, which adds at most the tree's depth (<= 56) to the list)
            uint32_t k = uni(n_nodes < n_workers ? n_nodes : n_workers);
            const uint32_t room = uni(n_nodes < kPoolNodeFull ? kPoolNodeFull - n_nodes : 1u);
            k = uni(k < room ? k : room);
            if (kCount && rank == 0)
                ++stats.wave_node_steps;
            // (everything inside ONE region of the k working lanes — ballots included, they only see those lanes: predicates that
            //  leave the region would travel as 0 / 1 words through vector registers)
            uint32_t next_nodes = 0, next_prims = 0;
            if (rank < k)
            {
                if (kCount)
                    stats.node_tests += 2;
                const uint32_t item = node_items[n_nodes - 1u - rank];
                const uint32_t ray_bits = item & ~kPoolMaxRef, r = item >> 10, node = item & kPoolMaxRef;
                const float4 a = rays[kRayVecs * r], b = rays[kRayVecs * r + 1];
                const uint32_t pack = __float_as_uint(b.w);
                const char *w = reinterpret_cast<const char *>(sc.walk_nodes) + 64u * node;
                const uint32_t ox = pack & 0xffu, oy = (pack >> 8) & 0xffu, oz = (pack >> 16) & 0xffu;
                const float *wnx = reinterpret_cast<const float *>(w + ox), *wfx = reinterpret_cast<const float *>(w + (16u - ox));
                const float *wny = reinterpret_cast<const float *>(w + oy), *wfy = reinterpret_cast<const float *>(w + (24u - oy));
                const float *wnz = reinterpret_cast<const float *>(w + oz), *wfz = reinterpret_cast<const float *>(w + (32u - oz));
                const float nx0 = (wnx[0] - a.x) * b.x, fx0 = (wfx[0] - a.x) * b.x;
                const float ny0 = (wny[0] - a.y) * b.y, fy0 = (wfy[0] - a.y) * b.y;
                const float nz0 = (wnz[0] - a.z) * b.z, fz0 = (wfz[0] - a.z) * b.z;
                const float nx1 = (wnx[8] - a.x) * b.x, fx1 = (wfx[8] - a.x) * b.x;
                const float ny1 = (wny[8] - a.y) * b.y, fy1 = (wfy[8] - a.y) * b.y;
                const float nz1 = (wnz[8] - a.z) * b.z, fz1 = (wfz[8] - a.z) * b.z;
                const float enter0 = fmaxf(fmaxf(fmaxf(kEpsDistance, nx0), ny0), nz0);
                const float enter1 = fmaxf(fmaxf(fmaxf(kEpsDistance, nx1), ny1), nz1);
                const bool hit0 = enter0 <= fminf(fminf(fminf(a.w, fx0), fy0), fz0);
                const bool hit1 = enter1 <= fminf(fminf(fminf(a.w, fx1), fy1), fz1);
                const uint32_t *refs = reinterpret_cast<const uint32_t *>(w);
                const uint32_t ref0 = refs[3], ref1 = refs[7];
                const bool first0 = enter0 <= enter1;
                const bool both = hit0 && hit1, some = hit0 || hit1;
                const uint32_t toward = (hit0 && (first0 || !hit1)) ? ref0 : ref1, other = first0 ? ref1 : ref0;
                const bool toward_leaf = static_cast<int32_t>(toward) < 0, other_leaf = static_cast<int32_t>(other) < 0; // kWalkLeaf = the sign bit
                // (ballots of plain comparisons, combined as 64-bit masks: a ballot of a combined predicate goes through a
                //  vector register as a 0 / 1 word)
                const unsigned long long b0 = __ballot(hit0), b1 = __ballot(hit1), b_tl = __ballot(toward_leaf), b_ol = __ballot(other_leaf);
                const unsigned long long b_some = b0 | b1, b_both = b0 & b1;
                const unsigned long long m_nl = b_some & b_tl, m_nn = b_some & ~b_tl, m_fl = b_both & b_ol, m_fn = b_both & ~b_ol;
                const uint32_t base_n = n_nodes - k, c_fn = static_cast<uint32_t>(__popcll(m_fn)), c_fl = static_cast<uint32_t>(__popcll(m_fl));
                // far children below near ones: the next step takes the near ones first
                if (both && !other_leaf)
                    (node_items + base_n)[pool_rank(m_fn)] = static_cast<uint16_t>(ray_bits | other);
                if (some && !toward_leaf)
                    (node_items + (base_n + c_fn))[pool_rank(m_nn)] = static_cast<uint16_t>(ray_bits | toward);
                if (both && other_leaf)
                    (prim_items + n_prims)[pool_rank(m_fl)] = static_cast<uint16_t>(ray_bits | (other & kPoolMaxRef));
                if (some && toward_leaf)
                    (prim_items + (n_prims + c_fl))[pool_rank(m_nl)] = static_cast<uint16_t>(ray_bits | (toward & kPoolMaxRef));
                next_nodes = base_n + c_fn + static_cast<uint32_t>(__popcll(m_nn));
                next_prims = n_prims + c_fl + static_cast<uint32_t>(__popcll(m_nl));
            }
            n_nodes = uni(next_nodes), n_prims = uni(next_prims); // (the first active lane has rank 0 < k: it took part)
            pool_sync();
        }
        if (kAny)
            break;
        // a ray whose candidate list overflowed walks again, its final bound as the initial one
        const bool again = pass == 0 && has_ray && counts[lane] > kPoolCands;
        const unsigned long long m_again = __ballot(again);
        if (m_again == 0)
            break;
        if (again)
        {
            counts[lane] = 0;
            node_items[pool_rank(m_again)] = static_cast<uint16_t>(lane << 10);
        }
        n_nodes = uni(static_cast<uint32_t>(__popcll(m_again)));
    }
    if (!has_ray)
        return false;
    if (kAny)
        return counts[lane] != 0;

    // ---- closest query: the owner decides among its candidates ----
    uint32_t n = counts[lane];
    if (n == 0)
        return false;
    n = n < kPoolCands ? n : kPoolCands;
    const uint2 *mine = cands + kPoolCands * lane;
    float t_min = __uint_as_float(mine[0].x);
    uint32_t winner = mine[0].y;
    for (uint32_t i = 1; i < n; ++i)
    {
        const float t = __uint_as_float(mine[i].x);
        if (t < t_min)
            t_min = t, winner = mine[i].y;
    }
    uint32_t n_tied = 0;
    for (uint32_t i = 0; i < n; ++i)
        n_tied += fabsf(__uint_as_float(mine[i].x) - t_min) <= tie ? 1u : 0u;
    if (n_tied > 1)
    {
        // rare: several hits within the tie radius of the nearest — the reference's own sequence on them, in rank order
        // (test_slot's pair rule, applied along the whole list)
        uint32_t last_rank = 0;
        float cur_t = 0.0f;
        for (uint32_t step = 0; step < n_tied; ++step)
        {
            uint32_t pick = 0, pick_rank = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < n; ++i)
            {
                if (!(fabsf(__uint_as_float(mine[i].x) - t_min) <= tie))
                    continue;
                const uint32_t rk = as_uint(sc.walk_prims[3 * static_cast<size_t>(mine[i].y) + 2].w);
                if ((step == 0 || rk > last_rank) && rk < pick_rank)
                    pick = i, pick_rank = rk;
            }
            const float t = __uint_as_float(mine[pick].x);
            const uint32_t slot = mine[pick].y;
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

#endif // __HIPCC__

} // namespace mcpt

#endif // MCPT_POOL_WALK_H
