// Multi-kernel wavefront formulation for gfx950 (MI355X), for scenes whose hierarchy lives in HBM.
//
// The single-kernel formulations (render_kernel_impl.h, stream_kernel_impl.h) run shading and ray queries in one
// kernel, so both get the register budget of the larger one (the shading code: 128 VGPRs = 4 wavefronts per SIMD, with
// spills) — and a hierarchy walk in HBM is bound by the latency of one node fetch per wavefront at a time (DESIGN.md
// section 6): what it needs is MORE wavefronts, not more lanes.  Here a frame is a sequence of rounds of two launches
// over path slots that live in HBM (one slot per pixel of the job; stream_core.h's slot storage):
//
//   shade   one lane per slot: fold in the shadow answers, finish / regenerate samples (a camera ray's hit comes from
//           the pre-pass, primary_kernel.hip), build the next vertex, emit up to 1 + S rays.  Emitted rays are
//           COMPACTED into the round's ray lists by wavefront ballot + prefix count, one global atomic per wavefront
//           and ray kind.  Neighbouring slots are neighbouring pixels, so a wavefront mostly shades one material.
//   trace   one lane per listed ray, nothing else in the kernel: walk state only (about 60 VGPRs, no spills), as many
//           wavefronts per SIMD as the traversal stacks in LDS allow — measured 3-4x the ray rate of the same walk
//           inside the render kernels (profiles/r02_experiments/lean_trace_kernel_rate.json).
//
// Same functions as the stream kernel (stream_shade, stream_load / stream_save, test_slot, walk_ordered_vote): the random
// stream of every pixel is consumed in the reference's order and the frame is the reference's bit for bit.
// Path state traffic: stream_cold_words + stream_hot_words per slot and round, coalesced (field-major arrays).
// Replaces the megakernel dispatch of the reference (src/renderer/renderer.cpp:88-95) for these scenes.
#include <hip/hip_runtime.h>

#include "../stream_core.h"
#include "render_kernel.h"

namespace mcpt
{

namespace
{

// Ray lists: kWavefrontQueues queues per ray kind, each with room for `region` ids (the slots of the workgroups that feed
// it); counters[parity][kind][queue] = ids listed this round.
constexpr uint32_t kWavefrontQueues = 8;

// counters[2 * parity] = extension rays listed this round, [2 * parity + 1] = shadow rays
#ifndef MCPT_WAVEFRONT_SHADE_WAVES
#define MCPT_WAVEFRONT_SHADE_WAVES 2
#endif
template <uint32_t kFeatures, uint32_t S>
__global__ void __launch_bounds__(kBlockSize, MCPT_WAVEFRONT_SHADE_WAVES) wavefront_shade(const DeviceScene sc, const RenderJob job, float *__restrict__ out,
                                                              uint32_t *__restrict__ cold, uint32_t *__restrict__ hot,
                                                              uint32_t *__restrict__ ids, uint32_t *__restrict__ counters, uint32_t n_slots,
                                                              uint32_t region, uint32_t first_round, uint32_t parity)
{
    using C = Config<kFeatures>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = i < n_slots;
    StreamStore m;
    m.hot = hot, m.cold = cold, m.P = n_slots;
    StreamSlot<S> s{};
    s.flags = kSlotExhausted;
    if (mine)
    {
        if (first_round)
        {
            // slot i renders item i of the job: its only pixel
            const uint32_t local_tile = i >> 6, r = i & 63u;
            const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
            const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);
            const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
            if (x < width && y < height)
            {
                s.flags = 0, s.item = i;
                start_pixel(s.st, y * width + x);
            }
        }
        else if (!(cold[kColdFlags * static_cast<size_t>(n_slots) + i] & kSlotExhausted)) // (a finished slot costs one word)
            stream_load<C, S>(m, i, s);
        if (!(s.flags & kSlotExhausted))
        {
        while (stream_shade<C, S>(sc, s, nullptr) == kStreamPixelDone)
        {
            const V3 c = pixel_value(sc, s.st);
            float *dst = out + 3 * static_cast<size_t>(job.packed ? s.item : s.st.pixel);
            dst[0] = c.x, dst[1] = c.y, dst[2] = c.z;
            s.flags |= kSlotExhausted;
        }
        stream_save<C, S>(m, i, s);
        }
    }
    // Compact the emitted rays into the round's lists: wavefront ballot + prefix count into the workgroup's LDS counters
    // (one LDS atomic per wavefront and kind), then ONE global atomic per workgroup and kind — on one of kWavefrontQueues
    // counters, chosen by the workgroup's index: 4096 workgroups hammering a single address cost more than the shading.
    __shared__ uint32_t wg_n[1 + S], wg_base[1 + S];
    if (threadIdx.x <= S)
        wg_n[threadIdx.x] = 0;
    __syncthreads();
    const bool live = mine && !(s.flags & kSlotExhausted);
    const bool ext = live && (s.flags & kSlotExtRay) != 0;
    const uint32_t at = wave_reserve(&wg_n[0], ext);
    uint32_t at_sh[S];
    bool sh[S];
#pragma unroll
    for (uint32_t k = 0; k < S; ++k)
    {
        sh[k] = live && (s.flags & (kSlotShadow0 << k)) != 0;
        at_sh[k] = wave_reserve(&wg_n[1], sh[k]);
    }
    __syncthreads();
    const uint32_t queue = blockIdx.x % kWavefrontQueues;
    if (threadIdx.x < 2)
        wg_base[threadIdx.x] = wg_n[threadIdx.x] ? atomicAdd(&counters[parity * 2 * kWavefrontQueues + threadIdx.x * kWavefrontQueues + queue], wg_n[threadIdx.x]) : 0u;
    __syncthreads();
    if (ext)
        ids[static_cast<size_t>(queue) * region + wg_base[0] + at] = i;
#pragma unroll
    for (uint32_t k = 0; k < S; ++k)
        if (sh[k])
            ids[(static_cast<size_t>(kWavefrontQueues) + queue) * region + wg_base[1] + at_sh[k]] = (1u + k) * n_slots + i;
}

template <bool kAnalytic, bool kSlivers>
__global__ void __launch_bounds__(kBlockSize) wavefront_trace(const DeviceScene sc, uint32_t *__restrict__ hot, const uint32_t *__restrict__ ids,
                                                              uint32_t *__restrict__ counters, uint32_t n_slots, uint32_t region, uint32_t parity)
{
    extern __shared__ uint32_t lds_stacks[];
    uint32_t *stack = lds_stacks + threadIdx.x;
    // workgroup -> (ray kind, queue, position in the queue): kinds are launched one after the other so that the long
    // closest-hit walks start first
    const uint32_t blocks_per_queue = region / kBlockSize;
    const uint32_t kind_of_list = blockIdx.x / (kWavefrontQueues * blocks_per_queue), rest = blockIdx.x % (kWavefrontQueues * blocks_per_queue);
    const uint32_t queue = rest % kWavefrontQueues, local = (rest / kWavefrontQueues) * kBlockSize + threadIdx.x;
    const uint32_t listed = counters[parity * 2 * kWavefrontQueues + kind_of_list * kWavefrontQueues + queue];
    if (blockIdx.x == 0 && threadIdx.x < 2 * kWavefrontQueues)
        counters[(parity ^ 1u) * 2 * kWavefrontQueues + threadIdx.x] = 0; // the next round's lists start empty
    if (local >= listed)
        return;
    const uint32_t P = n_slots;
    const uint32_t id = ids[(static_cast<size_t>(kind_of_list) * kWavefrontQueues + queue) * region + local];
    const uint32_t kind = id / P, slot = id - kind * P;
    TraceStats ts{0, 0, 0, 0};
    HitRaw hit;
    if (kind == 0)
    {
        const V3 o = stream_get3(hot, P, kHotA, slot), d = stream_get3(hot, P, kHotDir, slot);
        Ray ray = make_ray(o, d);
        const bool found = walk_ordered_vote<false, kAnalytic, false, kSlivers>(sc, stack, ray, hit, ts);
        hot[kHotPrim * static_cast<size_t>(P) + slot] = found ? hit.prim : kNone;
        if (found)
        {
            hot[kHotInst * static_cast<size_t>(P) + slot] = hit.inst | (hit.inside ? 0x80000000u : 0u);
            hot[kHotA * static_cast<size_t>(P) + slot] = as_uint(hit.a), hot[kHotB * static_cast<size_t>(P) + slot] = as_uint(hit.b);
            hot[kHotC * static_cast<size_t>(P) + slot] = as_uint(hit.c), hot[kHotT * static_cast<size_t>(P) + slot] = as_uint(ray.t_max);
        }
    }
    else
    {
        const uint32_t f = kHotShadow + 7u * (kind - 1u);
        const V3 o = stream_get3(hot, P, f, slot), d = stream_get3(hot, P, f + 3u, slot);
        Ray ray = make_ray(o, d);
        ray.t_max = as_float(hot[(f + 6u) * static_cast<size_t>(P) + slot]);
        if (walk_ordered_vote<true, kAnalytic, false, kSlivers>(sc, stack, ray, hit, ts))
            hot[(f + 6u) * static_cast<size_t>(P) + slot] = as_uint(-1.0f);
    }
}

constexpr uint32_t kSurfaceF = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
constexpr uint32_t kAllF = kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet;
constexpr uint32_t kWalkF = kFeatOrderedWalk | kFeatVoteWalk | kFeatSlivers;

} // namespace

bool WavefrontSupports(const DeviceScene &sc, const RenderJob &job)
{
    const uint32_t shadows = sc.integrator.n_emitters + (sc.integrator.n_area_lights ? 1u : 0u);
    // (instantiated for the surface-material feature set, one shadow ray per vertex: the mesh scenes)
    return !job.reference_walk && !sc.integrator.has_masks && sc.integrator.n_walk_nodes != 0 && shadows <= 1 &&
           (sc.features & ~kSurfaceF) == 0 && job.sample_split <= 1 && job.n_items != 0;
}

void WavefrontSizes(const DeviceScene &sc, uint32_t n_slots, size_t *cold_words, size_t *hot_words, size_t *id_words)
{
    *cold_words = size_t(stream_cold_words(1)) * n_slots, *hot_words = size_t(stream_hot_words(1)) * n_slots;
    const uint32_t blocks = (n_slots + kBlockSize - 1) / kBlockSize;
    const uint32_t region = ((blocks + kWavefrontQueues - 1) / kWavefrontQueues) * kBlockSize;
    *id_words = size_t(2) * kWavefrontQueues * region;
}

uint32_t WavefrontCounterWords() { return 4 * kWavefrontQueues; }

hipError_t LaunchWavefrontRound(const DeviceScene &sc, const RenderJob &job, float *out, uint32_t *cold, uint32_t *hot, uint32_t *ids,
                                uint32_t *counters, uint32_t n_slots, bool first_round, uint32_t parity, hipStream_t stream)
{
    const uint32_t blocks = (n_slots + kBlockSize - 1) / kBlockSize;
    const uint32_t region = ((blocks + kWavefrontQueues - 1) / kWavefrontQueues) * kBlockSize;
    hipLaunchKernelGGL((wavefront_shade<kSurfaceF | kWalkF, 1>), dim3(blocks), dim3(kBlockSize), 0, stream, sc, job, out, cold, hot, ids,
                       counters, n_slots, region, first_round ? 1u : 0u, parity);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess)
        return err;
    // (the ray count is known on the device only: the grid covers the most a round can list; empty workgroups leave at once)
    const size_t lds_bytes = size_t(sc.integrator.walk_depth) * kBlockSize * sizeof(uint32_t);
    hipLaunchKernelGGL((wavefront_trace<false, true>), dim3(2 * kWavefrontQueues * (region / kBlockSize)), dim3(kBlockSize), lds_bytes, stream,
                       sc, hot, ids, counters, n_slots, region, parity);
    return hipGetLastError();
}

} // namespace mcpt
