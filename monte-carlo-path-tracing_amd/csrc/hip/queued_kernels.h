// HIP kernels of the QUEUED renderer for gfx950 (MI355X): queue_core.h's formulation as launches.
//
// A frame = rounds.  Round r (parity p = r & 1):
//   trace   one lane per queued ray, walk state only (no path state, no shading code): extension rays first — the
//           answer goes, with the slot id, into the shade queue of the material group that was hit (wavefront ballot +
//           prefix count, ONE atomic per wavefront and group) — then shadow rays: an unoccluded ray adds its
//           contribution to the slot's radiance;
//   shade   one launch per material group present in the scene, each compiled with that group's BSDF model only:
//           one lane per queue entry, slot record in (96 B, array of structures), vertex, slot record out, emitted rays
//           compacted into the ray queues (ballot + prefix count, one atomic per wavefront and ray kind).
// Memory layout (all in HBM, sized for the job; 288 GB per GPU make one slot per pixel affordable up to 2^28 slots):
//   slots     n_slots x 96 B                          array of structures: a permuted wavefront touches <= 2 lines per slot
//   rays      kSubQueues x cap x (32 B | 48 B)        extension / shadow rays, written and read coalesced
//   entries   2 parities x groups x kSubQueues x cap x 32 B   shade queues: slot + hit record
//   counters  2 parities x (2 + groups) x kSubQueues, one per 128-byte line
// SUB-QUEUES: 64 independent queue sets.  Wavefront w of a launch works on sub-queue w % 64 and pushes into the same
// sub-queue, so a slot never leaves its sub-queue (capacity = slots / 64, exact) and the atomics of a launch go to 64
// (x groups) different lines instead of one — 16 384 wavefronts adding to ONE address serialise in L2 for longer than
// the shading takes.
// No MFMA: nothing here is a dense contraction.  Replaces the megakernel dispatch of the reference
// (src/renderer/renderer.cpp:88-95) and the per-hit material switch (src/renderer/bsdfs/bsdf.cpp:188-211).
#ifndef MCPT_QUEUED_KERNELS_H
#define MCPT_QUEUED_KERNELS_H

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../queue_core.h"
#include "render_kernel.h"

namespace mcpt
{

constexpr uint32_t kSubQueues = 64;
constexpr uint32_t kCounterStride = 32; // words: one counter per 128-byte line
constexpr uint32_t kQueueCounterKinds = 2 + kQueueGroups; // extension rays, shadow rays, one shade queue per group

struct QueueView
{
    uint32_t *slots, *rays_ext, *rays_shadow, *entries, *counters;
    uint32_t *spill;   // backing store of the trace launch's short traversal stacks (short_stack.h): walk_depth x lanes of the launch
    uint32_t cap;      // entries per sub-queue (a multiple of 64); slots = kSubQueues * cap
    uint32_t groups;   // bit g: the scene has vertices for group g's launch (bit 0 always)
    uint32_t n_present;
    uint32_t dense[kQueueGroups]; // group -> index among the present ones
};

__device__ __forceinline__ uint32_t *queue_counter(const QueueView &qv, uint32_t parity, uint32_t which, uint32_t q)
{
    return qv.counters + (static_cast<size_t>(parity * kQueueCounterKinds + which) * kSubQueues + q) * kCounterStride;
}
__device__ __forceinline__ uint32_t *queue_entries(const QueueView &qv, uint32_t parity, uint32_t group, uint32_t q)
{
    return qv.entries + ((static_cast<size_t>(parity) * qv.n_present + qv.dense[group]) * kSubQueues + q) * qv.cap * kQueueEntryWords;
}

// One shade-queue entry per lane where `p` holds, into the queue of that lane's group (uniform loop over the groups
// the scene has; one atomic per wavefront and group).
__device__ __forceinline__ void queue_push(const QueueView &qv, uint32_t parity, uint32_t q, bool p, uint32_t group, uint4 e0, uint4 e1)
{
#pragma unroll
    for (uint32_t g = 0; g < kQueueGroups; ++g)
    {
        if (!((qv.groups >> g) & 1u))
            continue;
        const bool here = p && group == g;
        if (__ballot(here) == 0)
            continue;
        const uint32_t pos = wave_reserve(queue_counter(qv, parity, 2 + g, q), here);
        if (here)
        {
            uint4 *dst = reinterpret_cast<uint4 *>(queue_entries(qv, parity, g, q) + static_cast<size_t>(pos) * kQueueEntryWords);
            dst[0] = e0, dst[1] = e1;
        }
    }
}

// kWaves: wavefronts per SIMD the launch is compiled for — 2 (256 VGPRs: no spills) or 4 (128 VGPRs)
template <uint32_t kFeatures, uint32_t kGroup, int kWaves>
__global__ void __launch_bounds__(kBlockSize, kWaves)
queued_shade(const DeviceScene sc, const RenderJob job, float *__restrict__ out, const QueueView qv, uint32_t parity, uint32_t fresh)
{
    using C = Config<kFeatures>;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlockSize + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlockSize / 64u);
    const uint32_t q = wave % kSubQueues, chunk_stride = n_waves / kSubQueues;
    const uint32_t listed = fresh ? qv.cap : *queue_counter(qv, parity, 2 + kGroup, q);
    const uint32_t *my_entries = queue_entries(qv, parity, kGroup, q);
    uint32_t *n_ext = queue_counter(qv, parity ^ 1u, 0, q), *n_shadow = queue_counter(qv, parity ^ 1u, 1, q);
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);
    const uint32_t n_slots = kSubQueues * qv.cap;
    // the slot takes work item `item`, or the next one nobody has taken yet, until one is a pixel of the film (edge
    // tiles are padded) or the job is exhausted
    auto assign = [&](StreamSlot<1> &s, uint32_t item)
    {
        for (;; item = n_slots + wave_reserve(job.work_counter, true))
        {
            if (item >= job.n_items)
            {
                s.flags = kSlotExhausted;
                return;
            }
            const uint32_t local_tile = job.tile_order ? static_cast<uint32_t>(job.tile_order[item >> 6]) : item >> 6, r = item & 63u;
            const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
            const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
            if (x < width && y < height)
            {
                s.flags = 0;
                start_pixel(s.st, y * width + x);
                return;
            }
        }
    };
    for (uint32_t c = wave / kSubQueues; c * 64u < listed; c += chunk_stride)
    {
        const uint32_t idx = c * 64u + lane;
        const bool mine = idx < listed;
        StreamSlot<1> s{};
        s.flags = kSlotExhausted;
        QueueRays rays{};
        uint32_t id = 0;
        uint4 *rec4 = nullptr;
        if (mine)
        {
            if (fresh)
            {
                id = q * qv.cap + idx; // slot i starts on work item i; the counter hands out the items behind the slots
                rec4 = reinterpret_cast<uint4 *>(qv.slots + static_cast<size_t>(id) * kQueueSlotWords);
                s.st.medium = kNone;
                assign(s, id);
            }
            else
            {
                const uint4 *e4 = reinterpret_cast<const uint4 *>(my_entries + static_cast<size_t>(idx) * kQueueEntryWords);
                const uint4 e0 = e4[0], e1 = e4[1];
                const uint32_t entry[kQueueEntryWords] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                id = entry[0];
                rec4 = reinterpret_cast<uint4 *>(qv.slots + static_cast<size_t>(id) * kQueueSlotWords);
                uint32_t rec[kQueueSlotWords];
                constexpr uint32_t kVecs = C::kVolPath ? 6u : 5u; // (words 18, 19 of a surface path's record are unused)
#pragma unroll
                for (uint32_t v = 0; v < kVecs; ++v)
                {
                    const uint4 t = rec4[v];
                    rec[4 * v] = t.x, rec[4 * v + 1] = t.y, rec[4 * v + 2] = t.z, rec[4 * v + 3] = t.w;
                }
                queue_load<C>(rec, s);
                queue_take_entry(entry, s);
            }
            uint32_t budget = kQueueSamplesPerCall;
            while (!(s.flags & kSlotExhausted) && queue_shade<C, kGroup>(sc, s, budget, rays, nullptr) == kQueuePixelDone)
            {
                const V3 v = pixel_value(sc, s.st);
                const size_t at = job.packed ? item_of_pixel(s.st.pixel, width, job.tiles_x, job.tile_first, job.tile_stride) : s.st.pixel;
                float *dst = out + 3 * at;
                dst[0] = v.x, dst[1] = v.y, dst[2] = v.z;
                assign(s, n_slots + wave_reserve(job.work_counter, true));
            }
            if (!(s.flags & kSlotExhausted))
            {
                uint32_t rec[kQueueSlotWords] = {};
                queue_save<C>(rec, s);
                constexpr uint32_t kVecs = C::kVolPath ? 6u : 5u;
#pragma unroll
                for (uint32_t v = 0; v < kVecs; ++v)
                    rec4[v] = uint4{rec[4 * v], rec[4 * v + 1], rec[4 * v + 2], rec[4 * v + 3]};
            }
        }
        // ---- emitted rays, compacted: wavefront ballot + prefix count, one atomic per wavefront and kind ----
        const bool live = mine && !(s.flags & kSlotExhausted);
        const bool ext = live && rays.ext, shadow = live && rays.shadow, requeue = live && rays.requeue;
        if (__ballot(ext))
        {
            const uint32_t pos = wave_reserve(n_ext, ext);
            if (ext)
            {
                uint4 *dst = reinterpret_cast<uint4 *>(qv.rays_ext + (static_cast<size_t>(q) * qv.cap + pos) * kQueueExtWords);
                const V3 o = s.st.origin, d = s.st.dir;
                dst[0] = uint4{as_uint(o.x), as_uint(o.y), as_uint(o.z), as_uint(d.x)};
                dst[1] = uint4{as_uint(d.y), as_uint(d.z), id, rays.miss_group};
            }
        }
        if (__ballot(shadow))
        {
            const uint32_t pos = wave_reserve(n_shadow, shadow);
            if (shadow)
            {
                uint4 *dst = reinterpret_cast<uint4 *>(qv.rays_shadow + (static_cast<size_t>(q) * qv.cap + pos) * kQueueShadowWords);
                const V3 o = s.sh_origin[0], d = s.sh_dir[0], k = rays.contribution;
                dst[0] = uint4{as_uint(o.x), as_uint(o.y), as_uint(o.z), as_uint(d.x)};
                dst[1] = uint4{as_uint(d.y), as_uint(d.z), as_uint(s.sh_tmax[0]), id | rays.shadow_id_bits};
                dst[2] = uint4{as_uint(k.x), as_uint(k.y), as_uint(k.z), 0u};
            }
        }
        if (__ballot(requeue))
            queue_push(qv, parity ^ 1u, q, requeue, rays.requeue_group, uint4{id, kQueueNoHit, 0u, 0u}, uint4{0u, 0u, 0u, 0u});
    }
}

// Grid of a persistent launch: every resident wavefront slot, a multiple of kSubQueues wavefronts.
template <class Kernel>
inline hipError_t QueuedGrid(Kernel kernel, size_t lds_bytes, uint32_t n_cus, uint32_t *blocks)
{
    int per_cu = 0;
    const hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlockSize, lds_bytes);
    if (err != hipSuccess)
        return err;
    if (per_cu < 1)
        return hipErrorOutOfMemory;
    constexpr uint32_t kBlocksPerSet = kSubQueues / (kBlockSize / 64u);
    *blocks = ((n_cus * static_cast<uint32_t>(per_cu) + kBlocksPerSet - 1) / kBlocksPerSet) * kBlocksPerSet;
    return hipSuccess;
}

constexpr uint32_t kQueuedFeatures = kFeatEmitters | kFeatTextures | kFeatMicrofacet | kFeatOrderedWalk | kFeatVoteWalk | kFeatSlivers;

template <uint32_t kGroup>
hipError_t LaunchQueuedShadeGroup(const DeviceScene &sc, const RenderJob &job, float *out, const QueueView &qv, uint32_t parity, bool fresh,
                                  uint32_t n_cus, hipStream_t stream)
{
    static thread_local uint32_t blocks = 0; // (per kernel instantiation)
    static const int waves = []
    {
        const char *e = std::getenv("MCPT_QUEUED_SHADE_WAVES"); // (measurements)
        return e && std::atoi(e) == 4 ? 4 : 2;
    }();
    if (blocks == 0)
    {
        const hipError_t err = waves == 4 ? QueuedGrid(queued_shade<kQueuedFeatures, kGroup, 4>, 0, n_cus, &blocks)
                                          : QueuedGrid(queued_shade<kQueuedFeatures, kGroup, 2>, 0, n_cus, &blocks);
        if (err != hipSuccess)
            return err;
    }
    if (waves == 4)
        hipLaunchKernelGGL((queued_shade<kQueuedFeatures, kGroup, 4>), dim3(blocks), dim3(kBlockSize), 0, stream, sc, job, out, qv, parity,
                           fresh ? 1u : 0u);
    else
        hipLaunchKernelGGL((queued_shade<kQueuedFeatures, kGroup, 2>), dim3(blocks), dim3(kBlockSize), 0, stream, sc, job, out, qv, parity,
                           fresh ? 1u : 0u);
    return hipGetLastError();
}

// one translation unit per group (hip/queued_shade_<g>.hip)
hipError_t LaunchQueuedShade0(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);
hipError_t LaunchQueuedShade1(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);
hipError_t LaunchQueuedShade2(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);
hipError_t LaunchQueuedShade3(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);
hipError_t LaunchQueuedShade4(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);
hipError_t LaunchQueuedShade5(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);
hipError_t LaunchQueuedShade6(const DeviceScene &, const RenderJob &, float *, const QueueView &, uint32_t, bool, uint32_t, hipStream_t);

} // namespace mcpt

#endif // MCPT_QUEUED_KERNELS_H
