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
#include <hip/hip_runtime.h>

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
    static constexpr int kWavesPerSimd = (kFeatures & (kFeatVolPath | kFeatAnalytic)) ? 3
#ifdef MCPT_EXPERIMENT_MICROFACET_WAVES
                                         : (kFeatures & kFeatMicrofacet)              ? MCPT_EXPERIMENT_MICROFACET_WAVES
#else
                                         : (kFeatures & kFeatMicrofacet)              ? 6
#endif
                                         : kLdsGeometry                                ? 4
                                                                                      : 6;
};

// kLdsGeometry: the arrays the ray queries and the light sampler read (both
// hierarchies, walk primitives, triangle positions) are copied into LDS by each
// workgroup before it starts and read from there (ds_read_b128) instead of
// through L1.  Used for scenes whose traversal data fits kLdsGeometryBytes
// (cornell: 8 KB); a walk is a chain of dependent loads, so the shorter LDS
// latency shortens every step.  Large scenes stream from HBM / L2.
// The ordered walk's stacks always live in LDS: lane t of the workgroup owns the
// words t, t + 256, t + 512, ... of the stack area.
template <uint32_t kFeatures, bool kCount, bool kLdsGeometry>
__global__ void __launch_bounds__(kBlockSize, (Budget<kFeatures, kLdsGeometry>::kWavesPerSimd))
render_kernel(const DeviceScene sc_in, const RenderJob job, float *__restrict__ out, TraceCounters *__restrict__ counters)
{
    using C = Config<kFeatures>;
    extern __shared__ float4 lds_geometry[];
    DeviceScene sc = sc_in;
    uint32_t n_staged = 0;
    if (kLdsGeometry)
    {
        const uint32_t n_node_vec = 2u * sc_in.integrator.n_nodes, n_tri_vec = 3u * sc_in.integrator.n_prims;
        const uint32_t n_walk_vec = C::kOrdered ? 4u * sc_in.integrator.n_walk_nodes : 0u;
        const uint32_t n_slot_vec = C::kOrdered ? n_tri_vec : 0u;
        for (uint32_t i = threadIdx.x; i < n_node_vec; i += blockDim.x)
            lds_geometry[i] = sc_in.nodes[i];
        for (uint32_t i = threadIdx.x; i < n_tri_vec; i += blockDim.x)
            lds_geometry[n_node_vec + i] = sc_in.tri_pos[i];
        for (uint32_t i = threadIdx.x; i < n_walk_vec; i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + i] = sc_in.walk_nodes[i];
        for (uint32_t i = threadIdx.x; i < n_slot_vec; i += blockDim.x)
            lds_geometry[n_node_vec + n_tri_vec + n_walk_vec + i] = sc_in.walk_prims[i];
        __syncthreads();
        sc.nodes = lds_geometry;
        sc.tri_pos = lds_geometry + n_node_vec;
        if (C::kOrdered)
        {
            sc.walk_nodes = lds_geometry + n_node_vec + n_tri_vec;
            sc.walk_prims = lds_geometry + n_node_vec + n_tri_vec + n_walk_vec;
        }
        n_staged = n_node_vec + n_tri_vec + n_walk_vec + n_slot_vec;
    }
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);

    LaneCounters local{};
    LaneCounters *cnt = kCount ? &local : nullptr;

    PathState st;
    st.alive = false;
    st.stack = reinterpret_cast<uint32_t *>(lds_geometry + n_staged) + threadIdx.x;
    bool has_pixel = false;
    uint32_t slot = 0; // where this pixel's result goes
    for (;;)
    {
        if (!has_pixel)
        {
            if (q >= job.n_items)
                break;
            // item -> tile -> pixel
            const uint32_t local_tile = q >> 6, r = q & 63u;
            const uint32_t tile = job.tile_first + local_tile * job.tile_stride;
            const uint32_t x = (tile % job.tiles_x) * 8u + (r & 7u), y = (tile / job.tiles_x) * 8u + (r >> 3);
            const uint32_t item = q;
            q += stride;
            if (x >= width || y >= height)
                continue; // padding of an edge tile
            const uint32_t pixel = y * width + x;
            start_pixel(st, pixel);
            slot = job.packed ? item : pixel;
            has_pixel = true;
        }
        if (!st.alive)
        {
            if (st.sample >= sc.camera.spp)
            {
                const V3 c = pixel_value(sc, st);
                float *dst = out + 3 * static_cast<size_t>(slot);
                dst[0] = c.x, dst[1] = c.y, dst[2] = c.z;
                has_pixel = false;
                continue;
            }
            start_sample(sc, st);
            if (kCount)
                ++local.samples;
        }
        path_step<C>(sc, st, cnt);
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
    }
}

// ---- unit kernels (diagnostics / parity tests): one query per lane ----------

// in: origin[3], dir[3] per ray.  out (19 floats per ray): valid, inside,
// instance, primitive-in-instance, t, uv[2], position, normal, tangent, bitangent.
template <bool kOrdered>
__global__ void intersect_kernel(const DeviceScene sc, uint32_t n, const float *__restrict__ rays,
                                 const uint32_t *__restrict__ seeds, float *__restrict__ out,
                                 uint32_t *__restrict__ seeds_out)
{
    extern __shared__ uint32_t lds_stack[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    Ray ray = make_ray(V3{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, V3{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
    uint32_t rng = seeds[i];
    HitRaw raw;
    TraceStats ts{0, 0, 0, 0};
    const bool hit = kOrdered ? walk_ordered<false, true, false>(sc, lds_stack + threadIdx.x, ray, raw, ts)
                              : walk_scene<false, true, true, false>(sc, ray, rng, raw, ts);
    float *o = out + 19 * static_cast<size_t>(i);
    for (int k = 0; k < 19; ++k)
        o[k] = 0.0f;
    o[2] = o[3] = -1.0f;
    o[4] = ray.t_max;
    seeds_out[i] = rng;
    if (!hit)
        return;
    const Surface s = make_surface<true, true>(sc, ray, raw);
    o[0] = 1.0f, o[1] = s.inside ? 1.0f : 0.0f, o[2] = static_cast<float>(s.inst);
    o[3] = static_cast<float>(raw.prim - sc.instances[s.inst].prim_base);
    o[5] = s.uv.u, o[6] = s.uv.v;
    const V3 v[4] = {s.position, s.normal, s.tangent, s.bitangent};
    for (int k = 0; k < 4; ++k)
        o[7 + 3 * k] = v[k].x, o[8 + 3 * k] = v[k].y, o[9 + 3 * k] = v[k].z;
}

// The steps of one pixel, one lane, reference-order walk (both walks give the same frame):
// per step 16 floats {ray origin[3], ray direction[3], throughput max, hit primitive (-1 = none),
// distance, shadow queries, last shadow result, LCG state after the step (bits), L[3] so far,
// depth}.  n_steps[0] = steps written.  The CPU build of the same code is tests/emu's
// mcpt_emu_debug_pixel: comparing the two shows where a device frame leaves the host's.
__global__ void trace_pixel_kernel(const DeviceScene sc, uint32_t pixel, uint32_t capacity, float *__restrict__ out,
                                   uint32_t *__restrict__ n_steps)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    PathState st;
    st.stack = nullptr;
    LaneCounters cnt{};
    start_pixel(st, pixel);
    uint32_t n = 0;
    while (!pixel_done(sc, st) && n < capacity)
    {
        if (!st.alive)
            start_sample(sc, st);
        path_step<Config<kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet>>(sc, st, &cnt);
        float *o = out + 16 * static_cast<size_t>(n++);
        for (int k = 0; k < 6; ++k)
            o[k] = cnt.last_closest_ray[k];
        o[6] = max_component(st.throughput);
        o[7] = cnt.last_hit_prim == kNone ? -1.0f : static_cast<float>(cnt.last_hit_prim);
        o[8] = cnt.last_hit_t, o[9] = static_cast<float>(cnt.last_shadow_count);
        o[10] = static_cast<float>(cnt.last_shadow_hit);
        o[11] = __uint_as_float(st.rng);
        o[12] = st.L.x, o[13] = st.L.y, o[14] = st.L.z, o[15] = static_cast<float>(st.depth);
    }
    n_steps[0] = n;
}

hipError_t LaunchTracePixel(const DeviceScene &sc, uint32_t pixel, uint32_t capacity, float *out, uint32_t *n_steps,
                            hipStream_t stream)
{
    hipLaunchKernelGGL(trace_pixel_kernel, dim3(1), dim3(64), 0, stream, sc, pixel, capacity, out, n_steps);
    return hipGetLastError();
}

// in (18 floats per query): wo, wi, normal, tangent, bitangent, uv, inside.
// out (8 floats): valid, pdf, attenuation[3], wi[3].  mode 0 = evaluate, 1 = sample.
__global__ void bsdf_kernel(const DeviceScene sc, uint32_t n, uint32_t id_bsdf, int mode,
                            const float *__restrict__ recs, const uint32_t *__restrict__ seeds,
                            float *__restrict__ out, uint32_t *__restrict__ seeds_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float *r = recs + 18 * static_cast<size_t>(i);
    BsdfQuery q;
    q.valid = false, q.pdf = 0, q.attenuation = V3{0, 0, 0};
    q.wo = V3{r[0], r[1], r[2]}, q.wi = V3{r[3], r[4], r[5]}, q.normal = V3{r[6], r[7], r[8]};
    q.tangent = V3{r[9], r[10], r[11]}, q.bitangent = V3{r[12], r[13], r[14]};
    q.uv = V2{r[15], r[16]}, q.inside = r[17] != 0.0f;
    uint32_t rng = seeds[i];
    const ShadeTables T = ShadeTables{sc.textures, sc.texels, sc.lut_brdf, sc.lut_albedo, false};
    if (mode == 0)
        bsdf_eval<true>(T, sc.bsdfs[id_bsdf], q);
    else
        bsdf_sample<true>(T, sc.bsdfs[id_bsdf], rng, q);
    float *o = out + 8 * static_cast<size_t>(i);
    o[0] = q.valid ? 1.0f : 0.0f, o[1] = q.pdf;
    o[2] = q.attenuation.x, o[3] = q.attenuation.y, o[4] = q.attenuation.z;
    o[5] = q.wi.x, o[6] = q.wi.y, o[7] = q.wi.z;
    seeds_out[i] = rng;
}

hipError_t LaunchIntersect(const DeviceScene &sc, uint32_t n, const float *rays, const uint32_t *seeds, float *out,
                           uint32_t *seeds_out, bool reference_walk, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    if (reference_walk || sc.integrator.has_masks)
        hipLaunchKernelGGL(intersect_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, rays, seeds,
                           out, seeds_out);
    else
        hipLaunchKernelGGL(intersect_kernel<true>, dim3((n + 255) / 256), dim3(256),
                           sc.integrator.walk_depth * 256 * sizeof(uint32_t), stream, sc, n, rays, seeds, out, seeds_out);
    return hipGetLastError();
}

hipError_t LaunchBsdf(const DeviceScene &sc, uint32_t n, uint32_t id_bsdf, int mode, const float *recs,
                      const uint32_t *seeds, float *out, uint32_t *seeds_out, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(bsdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, id_bsdf, mode, recs, seeds, out,
                       seeds_out);
    return hipGetLastError();
}

namespace
{

constexpr uint32_t kAll = kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet;

size_t StagedBytes(const DeviceScene &sc, bool ordered)
{
    size_t vecs = 2ull * sc.integrator.n_nodes + 3ull * sc.integrator.n_prims;
    if (ordered)
        vecs += 4ull * sc.integrator.n_walk_nodes + 3ull * sc.integrator.n_prims;
    return vecs * sizeof(float4);
}

template <uint32_t kFeatures, bool kCount, bool kLdsGeometry = false>
hipError_t Launch(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters, hipStream_t stream,
                  uint32_t max_blocks)
{
    constexpr bool kOrdered = (kFeatures & kFeatOrderedWalk) != 0;
    const size_t lds_bytes = (kLdsGeometry ? StagedBytes(sc, kOrdered) : 0) +
                             (kOrdered ? size_t(sc.integrator.walk_depth) * kBlockSize * sizeof(uint32_t) : 0);
    int per_cu = 0;
    hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &per_cu, render_kernel<kFeatures, kCount, kLdsGeometry>, kBlockSize, lds_bytes);
    if (err != hipSuccess)
        return err;
    if (per_cu < 1)
        per_cu = 1;
    uint32_t blocks = (job.n_items + kBlockSize - 1) / kBlockSize;
    const uint32_t resident = max_blocks * static_cast<uint32_t>(per_cu);
    if (blocks > resident)
        blocks = resident;
    if (blocks == 0)
        return hipSuccess;
    hipLaunchKernelGGL((render_kernel<kFeatures, kCount, kLdsGeometry>), dim3(blocks), dim3(kBlockSize), lds_bytes,
                       stream, sc, job, out, counters);
    return hipGetLastError();
}

} // namespace

// Picks the leanest instantiation that covers the scene's feature bits.  The
// ordered walk is the default; opacity masks (or job.reference_walk) select the
// reference-order walk, which exists only in the full instantiation.  The two
// LDS-resident instantiations (tiny scenes) use the plain ordered walk, all others
// the vote-scheduled one (its threshold comes from the commit: 0 for small scenes).
hipError_t LaunchRender(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters,
                        hipStream_t stream, uint32_t n_cus, const char **variant)
{
    constexpr uint32_t kO = kFeatOrderedWalk, kV = kFeatOrderedWalk | kFeatVoteWalk;
    const uint32_t f = sc.features;
    const bool ordered = !job.reference_walk && !sc.integrator.has_masks;
#if defined(MCPT_EXPERIMENT_LEAN_ONLY)
    // developer builds for kernel experiments: only the cornell instantiation (fast to compile)
    *variant = "diffuse-area+lds (experiment build)";
    if (counters != nullptr || !ordered || f != 0 || StagedBytes(sc, true) > kLdsGeometryBytes)
        return hipErrorNotSupported;
    return Launch<kO, false, true>(sc, job, out, nullptr, stream, n_cus);
#endif
    constexpr uint32_t kS = kFeatSlivers;
    const bool slivers = sc.integrator.walk_sliver_reach > 0.0f;
    if (counters != nullptr)
    {
        *variant = ordered ? "all+count" : "all+count, reference walk";
        return ordered ? Launch<kAll | kV | kS, true>(sc, job, out, counters, stream, n_cus)
                       : Launch<kAll, true>(sc, job, out, counters, stream, n_cus);
    }
    if (!ordered)
    {
        *variant = "all, reference walk";
        return Launch<kAll, false>(sc, job, out, nullptr, stream, n_cus);
    }
    constexpr uint32_t kSurface = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
    if (slivers)
    {
        // sliver triangles: the instantiations whose primitive test replays the reference on them
        if ((f & ~kSurface) == 0)
        {
            *variant = "surface-materials+slivers";
            return Launch<kSurface | kV | kS, false>(sc, job, out, nullptr, stream, n_cus);
        }
        *variant = "all+slivers";
        return Launch<kAll | kV | kS, false>(sc, job, out, nullptr, stream, n_cus);
    }
    const bool lds = StagedBytes(sc, true) <= kLdsGeometryBytes;
    if (f == 0)
    {
        *variant = lds ? "diffuse-area+lds" : "diffuse-area";
        return lds ? Launch<kO, false, true>(sc, job, out, nullptr, stream, n_cus)
                   : Launch<kV, false>(sc, job, out, nullptr, stream, n_cus);
    }
    if ((f & ~kFeatEmitters) == 0)
    {
        *variant = lds ? "diffuse-emitters+lds" : "diffuse-emitters";
        return lds ? Launch<kFeatEmitters | kO, false, true>(sc, job, out, nullptr, stream, n_cus)
                   : Launch<kFeatEmitters | kV, false>(sc, job, out, nullptr, stream, n_cus);
    }
    if ((f & ~kSurface) == 0)
    {
        *variant = "surface-materials";
        return Launch<kSurface | kV, false>(sc, job, out, nullptr, stream, n_cus);
    }
    *variant = "all";
    return Launch<kAll | kV, false>(sc, job, out, nullptr, stream, n_cus);
}

} // namespace mcpt
