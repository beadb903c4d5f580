// The ordered walk (traversal.h, walk_ordered_vote) with a SHORT traversal stack: the K most recent entries in LDS,
// anything older in a backing store in HBM.
//
// Why.  walk_ordered keeps a full stack per lane in LDS — one entry per level of the hierarchy (26-28 on the mesh
// configurations): 27 KiB per 256-lane workgroup, which caps a lean trace kernel at 5 wavefronts per SIMD although its
// registers would allow 8 — and the mesh walk is LATENCY bound: what it needs is more wavefronts in
// flight.  A near-child-first walk rarely holds more than a handful of postponed children at once, so the deep end of
// that stack is almost never touched.  Here a lane owns a ring of K entries in LDS (K = 8: 8 KiB per workgroup, 8
// workgroups per CU = 8 wavefronts per SIMD); when a ninth entry is pushed the OLDEST one moves to the lane's column of a
// backing store in global memory (a store, off the critical path), and it comes back (one load) only if the walk unwinds
// that far.  Same visiting order, same tests, same answers as walk_ordered_vote: only where an entry lives differs.
// north_star says "stackless": this is the bounded-state form of it that keeps the near-first order (a restart trail
// or parent links pay with re-descents / dependent loads on exactly the path that is latency bound).
#ifndef MCPT_SHORT_STACK_H
#define MCPT_SHORT_STACK_H

#include "traversal.h"

namespace mcpt
{

template <uint32_t K>
struct ShortStack
{
    static_assert((K & (K - 1u)) == 0, "ring size is a power of two");
    uint32_t *ring;        // this lane's ring: entry i at ring[(i & (K - 1)) * kWalkStackStride]
    uint32_t *spill;       // this lane's backing store: entry i at spill[i * spill_stride]
    uint32_t spill_stride; // (lanes of the launch: consecutive lanes, consecutive words)
    uint32_t base;         // entries [base, top] are in the ring, [0, base) in the backing store

    MCPT_HD void reset() { base = 0; }
    MCPT_HD void store(uint32_t i, uint32_t v)
    {
        if (i >= base + K) // (i == base + K: the walk pushes one entry at a time)
        {
            spill[static_cast<size_t>(base) * spill_stride] = ring[(base & (K - 1u)) * kWalkStackStride];
            ++base;
        }
        ring[(i & (K - 1u)) * kWalkStackStride] = v;
    }
    MCPT_HD uint32_t load(uint32_t i)
    {
        if (i < base) // unwound below the ring: the entry comes back from the backing store
        {
            const uint32_t v = spill[static_cast<size_t>(i) * spill_stride];
            ring[(i & (K - 1u)) * kWalkStackStride] = v;
            base = i;
            return v;
        }
        return ring[(i & (K - 1u)) * kWalkStackStride];
    }
};

// walk_ordered_vote, statement for statement, on a ShortStack.  (The node step's unconditional read of the entry below
// the top becomes a read only when it is needed — `none` — so that an unused read never touches the backing store.)
template <bool kAny, bool kAnalytic, bool kSlivers, uint32_t K>
MCPT_HD bool walk_ordered_short(const DeviceScene &sc, ShortStack<K> &stack, Ray &ray, HitRaw &hit)
{
    if (sc.integrator.n_walk_nodes == 0)
        return false;
    ClosestState best{false, ray.t_max, 0};
    stack.reset();
    stack.store(0, kWalkDone); // sentinel: reads as "no more work"
    uint32_t depth = 1;
    uint32_t cur = 0;
    const uint32_t break_below = sc.integrator.walk_break, hold_enough = sc.integrator.walk_hold;
    for (;;)
    {
        for (;;)
        {
            const bool searching = !(cur & kWalkLeaf);
            const uint32_t n_searching = lanes_where(searching);
            if (n_searching == 0)
                break;
            if (break_below | hold_enough)
            {
                const uint32_t n_holding = lanes_where(cur != kWalkDone && !searching);
                if ((n_searching < break_below && n_holding != 0) || (hold_enough && n_holding >= hold_enough))
                    break;
            }
            if (!searching)
                continue;
            const float4 *n = sc.walk_nodes + 4 * static_cast<size_t>(cur);
            const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            float enter0, enter1;
            const bool hit0 = box_enter(n0, n1, ray, enter0), hit1 = box_enter(n2, n3, ray, enter1);
            const uint32_t ref0 = as_uint(n0.w), ref1 = as_uint(n1.w);
            const bool first0 = enter0 <= enter1, both = hit0 && hit1, none = !(hit0 || hit1);
            const uint32_t toward = (hit0 && (first0 || !hit1)) ? ref0 : ref1;
            if (both)
                stack.store(depth, first0 ? ref1 : ref0);
            if (none)
                cur = stack.load(depth - 1);
            else
                cur = toward;
            depth = depth + (both ? 1u : 0u) - (none ? 1u : 0u);
        }
        if (lanes_where(cur != kWalkDone) == 0)
            break;
        if (cur == kWalkDone || !(cur & kWalkLeaf))
            continue;
        if (test_slot<kAny, kAnalytic, kSlivers>(sc, cur & ~kWalkLeaf, ray, hit, best) && kAny)
        {
            cur = kWalkDone;
            continue;
        }
        --depth;
        cur = stack.load(depth);
    }
    if (!kAny)
        ray.t_max = best.found ? best.best_t : ray.t_max;
    return best.found;
}

// ---- the 4-wide walk -------------------------------------------------------------------------
// The short stack of the calling lane over `ring` (kWideRing entries, lane-interleaved like every traversal stack) and
// the lane's column of the scene's backing store.
MCPT_HD ShortStack<kWideRing> wide_stack_of(const DeviceScene &sc, uint32_t *ring)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lanes = gridDim.x * blockDim.x, lane = blockIdx.x * blockDim.x + threadIdx.x;
    return ShortStack<kWideRing>{ring, sc.walk_spill + lane, lanes, 0u};
#else
    static thread_local uint32_t spill[3 * kWalkStackMax + 2];
    (void)sc;
    return ShortStack<kWideRing>{ring, spill, 1u, 0u};
#endif
}

MCPT_HD float wide_byte(uint32_t word, uint32_t i) { return static_cast<float>((word >> (8u * i)) & 0xFFu); }

// One node of the wide hierarchy against the ray: the four children's entry distances (kMaxFloat: not entered) and
// references.  Planes are decoded to world space — origin + scale * q, the operations the quantiser verified — and go
// through the slab test of box_enter: same formula as the exact hierarchy's, on boxes that contain the exact ones.
struct WideStep
{
    float enter[4];
    uint32_t ref[4];
};
MCPT_HD WideStep wide_node_test(const DeviceScene &sc, uint32_t node, const Ray &ray)
{
    const uint4 *n = sc.wide_nodes + 4 * static_cast<size_t>(node);
    const uint4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
    const V3 origin = V3{as_float(n0.x), as_float(n0.y), as_float(n0.z)};
    const float sx = as_float((n0.w & 0xFFu) << 23), sy = as_float(((n0.w >> 8) & 0xFFu) << 23), sz = as_float(((n0.w >> 16) & 0xFFu) << 23);
    const bool px = ray.dir_rcp.x > 0, py = ray.dir_rcp.y > 0, pz = ray.dir_rcp.z > 0;
    // the plane words the ray enters / leaves through, per axis (lo.x lo.y lo.z hi.x | hi.y hi.z)
    const uint32_t near_x = px ? n2.x : n2.w, far_x = px ? n2.w : n2.x;
    const uint32_t near_y = py ? n2.y : n3.x, far_y = py ? n3.x : n2.y;
    const uint32_t near_z = pz ? n2.z : n3.y, far_z = pz ? n3.y : n2.z;
    const uint32_t refs[4] = {n1.x, n1.y, n1.z, n1.w};
    WideStep out;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i)
    {
        const float nx = ((origin.x + sx * wide_byte(near_x, i)) - ray.origin.x) * ray.dir_rcp.x;
        const float fx = ((origin.x + sx * wide_byte(far_x, i)) - ray.origin.x) * ray.dir_rcp.x;
        const float ny = ((origin.y + sy * wide_byte(near_y, i)) - ray.origin.y) * ray.dir_rcp.y;
        const float fy = ((origin.y + sy * wide_byte(far_y, i)) - ray.origin.y) * ray.dir_rcp.y;
        const float nz = ((origin.z + sz * wide_byte(near_z, i)) - ray.origin.z) * ray.dir_rcp.z;
        const float fz = ((origin.z + sz * wide_byte(far_z, i)) - ray.origin.z) * ray.dir_rcp.z;
        const float t_enter = fmaxf(fmaxf(fmaxf(kEpsDistance, nx), ny), nz);
        const float t_exit = fminf(fminf(fminf(ray.t_max, fx), fy), fz);
        const bool entered = refs[i] != kWalkDone && t_enter <= t_exit;
        out.enter[i] = entered ? t_enter : kMaxFloat, out.ref[i] = entered ? refs[i] : kWalkDone;
    }
    return out;
}

// The children a ray entered, nearest first (a 5-exchange sorting network on (distance, reference); misses sort last);
// returns their number.
MCPT_HD uint32_t wide_sort(WideStep &w)
{
    auto order = [&](int a, int b)
    {
        const bool swap = w.enter[b] < w.enter[a];
        const float ea = w.enter[a], eb = w.enter[b];
        const uint32_t ra = w.ref[a], rb = w.ref[b];
        w.enter[a] = swap ? eb : ea, w.enter[b] = swap ? ea : eb;
        w.ref[a] = swap ? rb : ra, w.ref[b] = swap ? ra : rb;
    };
    order(0, 1), order(2, 3), order(0, 2), order(1, 3), order(1, 2);
    return (w.ref[0] != kWalkDone ? 1u : 0u) + (w.ref[1] != kWalkDone ? 1u : 0u) + (w.ref[2] != kWalkDone ? 1u : 0u) +
           (w.ref[3] != kWalkDone ? 1u : 0u);
}

// Node step of the wide walk for one lane: continue with the nearest entered child, postpone the others (farthest
// deepest), or take the most recently postponed reference when nothing was entered.
template <uint32_t K>
MCPT_HD void wide_node_step(const DeviceScene &sc, ShortStack<K> &stack, const Ray &ray, uint32_t &cur, uint32_t &depth)
{
    WideStep w = wide_node_test(sc, cur, ray);
    const uint32_t n = wide_sort(w);
    if (n == 0)
    {
        --depth;
        cur = stack.load(depth);
        return;
    }
    if (n > 3)
        stack.store(depth++, w.ref[3]);
    if (n > 2)
        stack.store(depth++, w.ref[2]);
    if (n > 1)
        stack.store(depth++, w.ref[1]);
    cur = w.ref[0];
}

// walk_ordered_vote's answers from the wide hierarchy: same scheduling (wavefront vote between node and primitive
// phases), same primitive test plus the explicit leaf-box check (test_slot, kLeafCheck).  `ring`: kWideRing entries
// per lane, lane-interleaved (LDS on the device).
template <bool kAny, bool kAnalytic, bool kCount, bool kSlivers = true>
MCPT_HD bool walk_wide_vote(const DeviceScene &sc, uint32_t *ring, Ray &ray, HitRaw &hit, TraceStats &stats)
{
    if (sc.integrator.n_wide_nodes == 0)
        return false;
    ShortStack<kWideRing> stack = wide_stack_of(sc, ring);
    ClosestState best{false, ray.t_max, 0};
    stack.store(0, kWalkDone);
    uint32_t depth = 1, cur = 0;
    const uint32_t break_below = sc.integrator.walk_break, hold_enough = sc.integrator.walk_hold;
    for (;;)
    {
        for (;;)
        {
            const bool searching = !(cur & kWalkLeaf);
            const uint32_t n_searching = lanes_where(searching);
            if (n_searching == 0)
                break;
            if (break_below | hold_enough)
            {
                const uint32_t n_holding = lanes_where(cur != kWalkDone && !searching);
                if ((n_searching < break_below && n_holding != 0) || (hold_enough && n_holding >= hold_enough))
                    break;
            }
            if (!searching)
                continue;
            if (kCount)
            {
                stats.node_tests += 2; // (one 64-byte record = two 32-byte units of SURVEY.md section 8(d), like a binary node with its two boxes)
                if (is_leading_lane())
                    ++stats.wave_node_steps;
            }
            wide_node_step(sc, stack, ray, cur, depth);
        }
        if (lanes_where(cur != kWalkDone) == 0)
            break;
        if (cur == kWalkDone || !(cur & kWalkLeaf))
            continue;
        if (kCount)
        {
            ++stats.prim_tests;
            if (is_leading_lane())
                ++stats.wave_prim_steps;
        }
        if (test_slot<kAny, kAnalytic, kSlivers, true>(sc, cur & ~kWalkLeaf, ray, hit, best) && kAny)
        {
            cur = kWalkDone;
            continue;
        }
        --depth;
        cur = stack.load(depth);
    }
    if (!kAny)
        ray.t_max = best.found ? best.best_t : ray.t_max;
    return best.found;
}

} // namespace mcpt

#endif // MCPT_SHORT_STACK_H
