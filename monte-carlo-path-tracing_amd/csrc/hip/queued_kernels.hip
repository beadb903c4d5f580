// Trace launch and host-side driver of the QUEUED renderer (queued_kernels.h, queue_core.h) for gfx950 (MI355X).
#include "queued_kernels.h"
#include "../short_stack.h"

#include <cstdlib>

namespace mcpt
{

namespace
{

// One lane per queued ray; a wavefront works through its sub-queue's extension rays, then its shadow rays.
// Walk state only: the ray, the best hit, and a SHORT traversal stack (short_stack.h: 8 entries per lane in LDS = 8 KiB
// per workgroup, older entries in HBM) — 8 wavefronts per SIMD where the full stacks allowed 5.
constexpr uint32_t kTraceRing = 8;
template <bool kAnalytic, bool kSlivers, int kWaves>
__global__ void __launch_bounds__(kBlockSize, kWaves) queued_trace(const DeviceScene sc, const QueueView qv, uint32_t parity)
{
    __shared__ uint32_t lds_rings[kTraceRing * kBlockSize];
    ShortStack<kTraceRing> stack;
    stack.ring = lds_rings + threadIdx.x;
    stack.spill = qv.spill + (static_cast<size_t>(blockIdx.x) * kBlockSize + threadIdx.x);
    stack.spill_stride = gridDim.x * kBlockSize;
    stack.base = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlockSize + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlockSize / 64u);
    const uint32_t q = wave % kSubQueues, chunk_stride = n_waves / kSubQueues;
    // the other parity's counters are free now (its queues were consumed by the previous round's launches) and are
    // filled next by this round's shade launches: zero them here
    if (blockIdx.x == 0)
        for (uint32_t k = threadIdx.x; k < kQueueCounterKinds * kSubQueues; k += kBlockSize)
            *queue_counter(qv, parity ^ 1u, k / kSubQueues, k % kSubQueues) = 0;
    // ---- extension rays: closest hit -> the shade queue of the material group that was hit ----
    const uint32_t n_ext = *queue_counter(qv, parity, 0, q);
    for (uint32_t c = wave / kSubQueues; c * 64u < n_ext; c += chunk_stride)
    {
        const uint32_t idx = c * 64u + lane;
        const bool mine = idx < n_ext;
        uint32_t id = 0, group = 0;
        uint4 e0{0, kNone, 0, 0}, e1{0, 0, 0, 0};
        if (mine)
        {
            const uint4 *r4 = reinterpret_cast<const uint4 *>(qv.rays_ext + (static_cast<size_t>(q) * qv.cap + idx) * kQueueExtWords);
            const uint4 r0 = r4[0], r1 = r4[1];
            id = r1.z;
            Ray ray = make_ray(V3{as_float(r0.x), as_float(r0.y), as_float(r0.z)}, V3{as_float(r0.w), as_float(r1.x), as_float(r1.y)});
            HitRaw hit;
            const bool found = walk_ordered_short<false, kAnalytic, kSlivers, kTraceRing>(sc, stack, ray, hit);
            e0.x = id;
            group = r1.w; // where the slot goes if the ray left the scene
            if (found)
            {
                group = queue_group_of_instance(sc, hit.inst);
                e0.y = hit.prim, e0.z = hit.inst | (hit.inside ? 0x80000000u : 0u), e0.w = as_uint(hit.a);
                e1.x = as_uint(hit.b), e1.y = as_uint(hit.c), e1.z = as_uint(ray.t_max);
            }
        }
        queue_push(qv, parity, q, mine, group, e0, e1);
    }
    // ---- shadow rays: any hit.  Unoccluded: the slot's radiance gains the ray's contribution ----
    // (chunks dealt from the LAST wavefront of the sub-queue down: the wavefronts that got extension rays above are the
    //  first ones)
    const uint32_t n_shadow = *queue_counter(qv, parity, 1, q);
    for (uint32_t c = chunk_stride - 1u - wave / kSubQueues; c * 64u < n_shadow; c += chunk_stride)
    {
        const uint32_t idx = c * 64u + lane;
        const bool mine = idx < n_shadow;
        uint32_t id = 0;
        if (mine)
        {
            const uint4 *r4 = reinterpret_cast<const uint4 *>(qv.rays_shadow + (static_cast<size_t>(q) * qv.cap + idx) * kQueueShadowWords);
            const uint4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
            id = r1.w;
            Ray ray = make_ray(V3{as_float(r0.x), as_float(r0.y), as_float(r0.z)}, V3{as_float(r0.w), as_float(r1.x), as_float(r1.y)});
            ray.t_max = as_float(r1.z);
            HitRaw hit;
            if (!walk_ordered_short<true, kAnalytic, kSlivers, kTraceRing>(sc, stack, ray, hit))
            {
                // L += throughput * direct (stream_fold / connect_lights): nobody else touches this slot's radiance
                // during the launch
                float *L = reinterpret_cast<float *>(qv.slots + static_cast<size_t>(id & kQueueSlotMask) * kQueueSlotWords + kQL);
                L[0] = L[0] + as_float(r2.x), L[1] = L[1] + as_float(r2.y), L[2] = L[2] + as_float(r2.z);
            }
        }
        const bool push = mine && (id & kQueuePush) != 0; // the slot has no extension ray out: queue it for its next launch
        if (__ballot(push))
            queue_push(qv, parity, q, push, (id >> kQueueGroupShift) & 7u, uint4{id & kQueueSlotMask, kQueueNoHit, 0u, 0u},
                       uint4{0u, 0u, 0u, 0u});
    }
}

// Wavefronts per SIMD the trace launch is compiled for and launched with: 8 (64 VGPRs, a few spills), 6 (80) or 5 (its
// natural 88).  MCPT_QUEUED_TRACE_WAVES in the environment picks another one for measurements.
int TraceWaves()
{
    static const int waves = []
    {
        const char *e = std::getenv("MCPT_QUEUED_TRACE_WAVES");
        const int w = e ? std::atoi(e) : 8;
        return w == 5 || w == 6 ? w : 8;
    }();
    return waves;
}

template <bool kAnalytic, bool kSlivers>
hipError_t LaunchTrace(const DeviceScene &sc, const QueueView &qv, uint32_t parity, uint32_t n_cus, hipStream_t stream)
{
    const uint32_t blocks = QueuedTraceBlocks(n_cus);
    switch (TraceWaves())
    {
    case 5: hipLaunchKernelGGL((queued_trace<kAnalytic, kSlivers, 5>), dim3(blocks), dim3(kBlockSize), 0, stream, sc, qv, parity); break;
    case 6: hipLaunchKernelGGL((queued_trace<kAnalytic, kSlivers, 6>), dim3(blocks), dim3(kBlockSize), 0, stream, sc, qv, parity); break;
    default: hipLaunchKernelGGL((queued_trace<kAnalytic, kSlivers, 8>), dim3(blocks), dim3(kBlockSize), 0, stream, sc, qv, parity); break;
    }
    return hipGetLastError();
}

} // namespace

// Grid of the trace launch: every wavefront slot of the machine at 8 per SIMD, a multiple of kSubQueues wavefronts.
uint32_t QueuedTraceBlocks(uint32_t n_cus)
{
    constexpr uint32_t kBlocksPerSet = kSubQueues / (kBlockSize / 64u);
    return ((n_cus * static_cast<uint32_t>(TraceWaves()) + kBlocksPerSet - 1) / kBlocksPerSet) * kBlocksPerSet;
}

bool QueuedSupports(const DeviceScene &sc, const RenderJob &job)
{
    const uint32_t shadows = sc.integrator.n_emitters + (sc.integrator.n_area_lights ? 1u : 0u);
    constexpr uint32_t kSceneF = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
    // surface paths on triangle meshes, one shadow ray per vertex, camera rays from the pre-pass
    return PrimaryPrepassSupports(sc, job) && shadows <= 1 && (sc.features & ~kSceneF) == 0 && job.sample_split <= 1 &&
           !job.independent_samples && job.n_items != 0;
}

uint32_t QueuedGroups(const BsdfRec *bsdfs, size_t n_bsdfs, bool any_instance_without_bsdf)
{
    uint32_t groups = 1u; // (misses, emitters, pass-through surfaces)
    (void)any_instance_without_bsdf;
    for (size_t k = 0; k < n_bsdfs; ++k)
        groups |= 1u << queue_group_of_kind(bsdfs[k].kind);
    return groups;
}

void QueuedLayout(uint32_t n_slots_wanted, uint32_t groups, uint32_t walk_depth, uint32_t n_cus, QueuedSizes *sz)
{
    sz->spill_words = size_t(walk_depth) * QueuedTraceBlocks(n_cus) * kBlockSize;
    uint32_t cap = (n_slots_wanted + kSubQueues - 1) / kSubQueues;
    cap = ((cap + 63u) / 64u) * 64u;
    if (cap == 0)
        cap = 64;
    uint32_t present = 0;
    for (uint32_t g = 0; g < kQueueGroups; ++g)
        present += (groups >> g) & 1u;
    sz->cap = cap, sz->n_slots = cap * kSubQueues, sz->n_present = present;
    sz->slot_words = size_t(sz->n_slots) * kQueueSlotWords;
    sz->ext_words = size_t(sz->n_slots) * kQueueExtWords;
    sz->shadow_words = size_t(sz->n_slots) * kQueueShadowWords;
    sz->entry_words = size_t(2) * present * sz->n_slots * kQueueEntryWords;
    sz->counter_words = size_t(2) * kQueueCounterKinds * kSubQueues * kCounterStride;
}

static QueueView MakeView(uint32_t *base, const QueuedSizes &sz, uint32_t groups)
{
    QueueView qv{};
    qv.slots = base;
    qv.rays_ext = qv.slots + sz.slot_words;
    qv.rays_shadow = qv.rays_ext + sz.ext_words;
    qv.entries = qv.rays_shadow + sz.shadow_words;
    qv.counters = qv.entries + sz.entry_words;
    qv.spill = qv.counters + sz.counter_words;
    qv.cap = sz.cap, qv.groups = groups, qv.n_present = sz.n_present;
    uint32_t d = 0;
    for (uint32_t g = 0; g < kQueueGroups; ++g)
        qv.dense[g] = ((groups >> g) & 1u) ? d++ : 0u;
    return qv;
}

uint32_t *QueuedCounters(uint32_t *base, const QueuedSizes &sz) { return MakeView(base, sz, 1u).counters; }

hipError_t LaunchQueuedRound(const DeviceScene &sc, const RenderJob &job, float *out, uint32_t *base, const QueuedSizes &sz, uint32_t groups,
                             uint32_t round, uint32_t n_cus, hipStream_t stream)
{
    const QueueView qv = MakeView(base, sz, groups);
    const uint32_t parity = round & 1u;
    using ShadeFn = hipError_t (*)(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);
    static const ShadeFn kShade[kQueueGroups] = {LaunchQueuedShade0, LaunchQueuedShade1, LaunchQueuedShade2, LaunchQueuedShade3,
                                                 LaunchQueuedShade4, LaunchQueuedShade5, LaunchQueuedShade6};
    if (round == 0) // every slot takes its first pixel and starts its first sample
        return kShade[0](sc, job, out, qv, 0, true, n_cus, stream);
    const bool slivers = sc.integrator.walk_sliver_reach > 0.0f;
    hipError_t err = slivers ? LaunchTrace<false, true>(sc, qv, parity, n_cus, stream) : LaunchTrace<false, false>(sc, qv, parity, n_cus, stream);
    for (uint32_t g = 0; g < kQueueGroups && err == hipSuccess; ++g)
        if ((groups >> g) & 1u)
            err = kShade[g](sc, job, out, qv, parity, false, n_cus, stream);
    return err;
}

} // namespace mcpt
