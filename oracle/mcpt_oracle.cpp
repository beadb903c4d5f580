// TEST INFRASTRUCTURE — CPU oracle for the render hot path.  NOT shipped, NOT
// linked into the product: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library, and only as the checker.
//
// What it is: a plain scalar C++ restatement of the reference CPU integrator
// (zhiwei-c/Monte-Carlo-Path-Tracing, C++17) for the path BASELINE.json's
// north_star names: per-pixel sample loop (reference
// src/renderer/renderer.cpp:62-85) -> Integrator::Shade (integrators/*.cpp)
// -> two-level BVH traversal (rtcore/*) -> BSDF / emitter / medium code.
// Every function cites the reference file:line it follows
// (oracle_math.hpp, oracle_scene.hpp, oracle_shading.hpp, oracle_trace.hpp,
// oracle_integrators.hpp).
//
// Parity status: PINNED.  tests/test_oracle_vs_reference.py compares this
// library bit-for-bit with the real reference compiled from its own sources
// (oracle/_ref, built by oracle/Makefile in the authoring container) and
// tests/test_oracle_golden.py compares it with committed frames the compiled
// reference produced (tests/golden/, generator tests/golden/make_golden.py).
//
// Threading: like the reference's DispathRaysCpu (renderer.cpp:142-253)
// pixels are independent (one LCG state per pixel, seeded from the pixel
// index) so worker threads only partition pixels; the image does not depend
// on the thread count.
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>

#include "oracle_integrators.hpp"

namespace
{

using namespace orc;

thread_local std::string g_error;

// renderer.cpp:62-85
void RenderPixel(const Scene &sc, uint32_t i, uint32_t j, float *frame, TraceStats *st)
{
    const Camera &cam = sc.camera;
    const uint32_t offset = (j * cam.width + i) * 3;
    uint32_t rng = Tea4(offset, 0);
    V3 sum;
    for (uint32_t s = 0; s < cam.spp; ++s)
    {
        const float u = s * cam.spp_inv, v = RadicalInverse<2>(s + 1),
                    x = 2.0f * (i + u) / cam.width - 1.0f,
                    y = 1.0f - 2.0f * (j + v) / cam.height;
        const V3 look = Unit(cam.front + x * cam.dx + y * cam.dy);
        V3 c = sc.volpath ? ShadeVolPath(sc, cam.eye, look, &rng, st)
                          : ShadePath(sc, cam.eye, look, &rng, st);
        c.x = fminf(c.x, 1.0f); // per-sample clamp before averaging (quirk Q3)
        c.y = fminf(c.y, 1.0f);
        c.z = fminf(c.z, 1.0f);
        sum += c;
    }
    sum *= cam.spp_inv;
    frame[offset + 0] = sum.x, frame[offset + 1] = sum.y, frame[offset + 2] = sum.z;
}

struct Handle
{
    Scene scene;
};

void RenderRange(const Scene &sc, float *frame, uint32_t first_pixel, uint32_t n_pixel,
                 int n_threads, TraceStats *total)
{
    if (n_threads <= 0)
        n_threads = static_cast<int>(std::thread::hardware_concurrency());
    if (n_threads < 1)
        n_threads = 1;
    std::atomic<uint32_t> next{0};
    constexpr uint32_t kChunk = 64; // same granularity as renderer.cpp:42
    std::vector<TraceStats> stats(n_threads);
    auto work = [&](int tid)
    {
        for (;;)
        {
            const uint32_t begin = next.fetch_add(kChunk);
            if (begin >= n_pixel)
                break;
            const uint32_t end = std::min(begin + kChunk, n_pixel);
            for (uint32_t p = first_pixel + begin; p < first_pixel + end; ++p)
                RenderPixel(sc, p % sc.camera.width, p / sc.camera.width, frame,
                            total ? &stats[tid] : nullptr);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t)
        pool.emplace_back(work, t);
    work(0);
    for (std::thread &t : pool)
        t.join();
    if (total)
        for (const TraceStats &s : stats)
        {
            total->closest_rays += s.closest_rays, total->shadow_rays += s.shadow_rays;
            total->node_tests += s.node_tests, total->prim_tests += s.prim_tests;
        }
}

} // namespace

extern "C"
{

const char *mcpt_oracle_last_error(void) { return g_error.c_str(); }

// Loads and commits an MCSD scene.  Returns nullptr on error.
void *mcpt_oracle_load(const char *mcsd_path)
{
    try
    {
        Handle *h = new Handle;
        CommitScene(mcsd::Load(mcsd_path), &h->scene);
        return h;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return nullptr;
    }
}

void mcpt_oracle_free(void *handle) { delete static_cast<Handle *>(handle); }

void mcpt_oracle_dims(void *handle, int *width, int *height, uint32_t *spp)
{
    const Scene &sc = static_cast<Handle *>(handle)->scene;
    *width = sc.camera.width, *height = sc.camera.height, *spp = sc.camera.spp;
}

// Renders pixels [first_pixel, first_pixel + n_pixel) (row-major pixel
// index) into the full-size frame buffer (width*height*3 float32, row 0 =
// top).  n_threads <= 0 -> hardware_concurrency().  `seconds` gets the wall
// time of the render loop; `stats` (4 x u64: closest rays, shadow rays, node
// tests, primitive tests) may be NULL.
int mcpt_oracle_render(void *handle, float *frame, uint32_t first_pixel,
                       uint32_t n_pixel, int n_threads, double *seconds,
                       uint64_t *stats)
{
    try
    {
        const Scene &sc = static_cast<Handle *>(handle)->scene;
        const uint32_t total = static_cast<uint32_t>(sc.camera.width) * sc.camera.height;
        if (first_pixel > total || n_pixel > total - first_pixel)
            throw std::runtime_error("pixel range outside the frame");
        TraceStats st;
        const auto t0 = std::chrono::steady_clock::now();
        RenderRange(sc, frame, first_pixel, n_pixel, n_threads, stats ? &st : nullptr);
        const auto t1 = std::chrono::steady_clock::now();
        if (seconds)
            *seconds = std::chrono::duration<double>(t1 - t0).count();
        if (stats)
        {
            stats[0] = st.closest_rays, stats[1] = st.shadow_rays;
            stats[2] = st.node_tests, stats[3] = st.prim_tests;
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// Per-sample trace of one pixel (same loop as RenderPixel): unclamped
// radiance of every sample and the LCG state after it.
void mcpt_oracle_trace_pixel(void *handle, uint32_t i, uint32_t j,
                             float *radiance, uint32_t *state_after)
{
    const Scene &sc = static_cast<Handle *>(handle)->scene;
    const Camera &cam = sc.camera;
    uint32_t rng = Tea4((j * cam.width + i) * 3, 0);
    for (uint32_t s = 0; s < cam.spp; ++s)
    {
        const float u = s * cam.spp_inv, v = RadicalInverse<2>(s + 1),
                    x = 2.0f * (i + u) / cam.width - 1.0f,
                    y = 1.0f - 2.0f * (j + v) / cam.height;
        const V3 look = Unit(cam.front + x * cam.dx + y * cam.dy);
        const V3 c = sc.volpath ? ShadeVolPath(sc, cam.eye, look, &rng)
                                : ShadePath(sc, cam.eye, look, &rng);
        radiance[3 * s + 0] = c.x, radiance[3 * s + 1] = c.y, radiance[3 * s + 2] = c.z;
        state_after[s] = rng;
    }
}

// BSDF unit call; same record layout as mcpt_ref_bsdf in ref_driver.cpp.
void mcpt_oracle_bsdf(void *handle, uint32_t id_bsdf, int mode, const float *in,
                      uint32_t *state, float *out)
{
    const Scene &sc = static_cast<Handle *>(handle)->scene;
    Scatter r;
    r.wo = {in[0], in[1], in[2]}, r.wi = {in[3], in[4], in[5]};
    r.normal = {in[6], in[7], in[8]}, r.tangent = {in[9], in[10], in[11]};
    r.bitangent = {in[12], in[13], in[14]};
    r.uv = {in[15], in[16]};
    r.inside = in[17] != 0.0f;
    if (mode == 0)
        BsdfEval(sc, sc.bsdfs[id_bsdf], &r);
    else
        BsdfSample(sc, sc.bsdfs[id_bsdf], state, &r);
    out[0] = r.valid ? 1.0f : 0.0f, out[1] = r.pdf;
    out[2] = r.attenuation.x, out[3] = r.attenuation.y, out[4] = r.attenuation.z;
    out[5] = r.wi.x, out[6] = r.wi.y, out[7] = r.wi.z;
}

// Closest-hit unit call; same output layout as mcpt_ref_intersect.
void mcpt_oracle_intersect(void *handle, const float *origin, const float *dir,
                           uint32_t *state, float *out)
{
    const Scene &sc = static_cast<Handle *>(handle)->scene;
    Ray ray = MakeRay({origin[0], origin[1], origin[2]}, {dir[0], dir[1], dir[2]});
    const Hit hit = ClosestHit(sc, state, &ray);
    out[0] = hit.valid, out[1] = hit.inside;
    out[2] = hit.inst == kNone ? -1.0f : static_cast<float>(hit.inst);
    out[3] = hit.prim == kNone ? -1.0f : static_cast<float>(hit.prim);
    out[4] = ray.t_max, out[5] = hit.uv.u, out[6] = hit.uv.v;
    const V3 v[4] = {hit.position, hit.normal, hit.tangent, hit.bitangent};
    for (int k = 0; k < 4; ++k)
        out[7 + 3 * k] = v[k].x, out[8 + 3 * k] = v[k].y, out[9 + 3 * k] = v[k].z;
}

// ---- table accessors for the builder parity tests -------------------------

uint32_t mcpt_oracle_node_count(void *handle)
{
    return static_cast<uint32_t>(static_cast<Handle *>(handle)->scene.nodes.size());
}

// Per node: leaf,left,right,object (u32 x4) and area, box lo xyz, hi xyz (f32 x7).
void mcpt_oracle_nodes(void *handle, uint32_t *links, float *geom)
{
    const Scene &sc = static_cast<Handle *>(handle)->scene;
    for (size_t i = 0; i < sc.nodes.size(); ++i)
    {
        const Node &n = sc.nodes[i];
        links[4 * i + 0] = n.leaf ? 1u : 0u, links[4 * i + 1] = n.left;
        links[4 * i + 2] = n.right, links[4 * i + 3] = n.object;
        geom[7 * i + 0] = n.area;
        geom[7 * i + 1] = n.box.lo.x, geom[7 * i + 2] = n.box.lo.y, geom[7 * i + 3] = n.box.lo.z;
        geom[7 * i + 4] = n.box.hi.x, geom[7 * i + 5] = n.box.hi.y, geom[7 * i + 6] = n.box.hi.z;
    }
}

void mcpt_oracle_kulla_conty(float *brdf, float *albedo)
{
    std::vector<float> b, a;
    BuildKullaConty(&b, &a);
    std::memcpy(brdf, b.data(), b.size() * sizeof(float));
    std::memcpy(albedo, a.data(), a.size() * sizeof(float));
}

int mcpt_oracle_bvh_build(uint32_t n, const float *aabbs, const float *areas,
                          uint32_t *leaf, uint32_t *left, uint32_t *right,
                          uint32_t *object, float *area, float *box)
{
    std::vector<Box> boxes(n);
    std::vector<float> ar(areas, areas + n);
    for (uint32_t i = 0; i < n; ++i)
    {
        boxes[i].lo = {aabbs[6 * i], aabbs[6 * i + 1], aabbs[6 * i + 2]};
        boxes[i].hi = {aabbs[6 * i + 3], aabbs[6 * i + 4], aabbs[6 * i + 5]};
    }
    const std::vector<Node> nodes = lbvh::Build(boxes, ar);
    for (size_t i = 0; i < nodes.size(); ++i)
    {
        leaf[i] = nodes[i].leaf ? 1u : 0u;
        left[i] = nodes[i].left, right[i] = nodes[i].right, object[i] = nodes[i].object;
        area[i] = nodes[i].area;
        box[6 * i + 0] = nodes[i].box.lo.x, box[6 * i + 1] = nodes[i].box.lo.y, box[6 * i + 2] = nodes[i].box.lo.z;
        box[6 * i + 3] = nodes[i].box.hi.x, box[6 * i + 4] = nodes[i].box.hi.y, box[6 * i + 5] = nodes[i].box.hi.z;
    }
    return static_cast<int>(nodes.size());
}

uint32_t mcpt_oracle_tea4(uint32_t v0, uint32_t v1) { return Tea4(v0, v1); }
float mcpt_oracle_random_float(uint32_t *state) { return Rand(state); }
float mcpt_oracle_vdc2(uint32_t i) { return RadicalInverse<2>(i); }
float mcpt_oracle_vdc3(uint32_t i) { return RadicalInverse<3>(i); }

} // extern "C"
