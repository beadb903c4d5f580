// TEST INFRASTRUCTURE — CPU oracle, part 2: committed scene tables.
//
// Restates the host-side builders whose OUTPUT the hot path consumes
// (SURVEY.md §8 row a21): mesh baking and tangent frames
// (reference src/rtcore/scene.cpp:15-111,196-324), analytic shape set-up
// (scene.cpp:326-472), the Morton LBVH (src/rtcore/accel/bvh_builder.cpp:
// 74-207), the two-level node array with the TLAS first (scene.cpp:474-533),
// light tables (src/renderer/renderer.cpp:271-304), BSDF constants
// (src/renderer/bsdfs/bsdf.cpp:112-186), medium constants
// (src/renderer/medium/medium.cpp:6-39), env-map tables
// (src/renderer/emitters/envmap.cpp:20-68 + renderer.cpp:597-606 +
// emitter.cpp:166-175), the Kulla-Conty LUT (kulla_conty.cpp:12-80) and the
// camera frame (src/renderer/camera.cpp:26-37).
//
// Storage is index-based flat tables instead of the reference's pointer
// graphs; the arithmetic and the table contents are the same.
#ifndef ORACLE_SCENE_HPP
#define ORACLE_SCENE_HPP

#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "mcsd_scene.hpp"
#include "oracle_math.hpp"

namespace orc
{

struct Box
{
    V3 lo = V3(kMaxF), hi = V3(kLowestF); // aabb.cpp:8
    void Grow(V3 p) { lo = Min3(p, lo), hi = Max3(p, hi); }
    void Grow(const Box &b) { lo = Min3(b.lo, lo), hi = Max3(b.hi, hi); }
    V3 Center() const { return (lo + hi) * 0.5f; }
};

struct Node // bvh_builder.hpp:11-25
{
    bool leaf = true;
    uint32_t left = kNone, right = kNone, object = kNone;
    float area = 0;
    Box box;
};

enum class Shape : uint32_t { kTriangle = 1, kSphere, kDisk, kCylinder };

struct Prim // primitive.hpp:25-55 (tagged union flattened)
{
    Shape shape = Shape::kTriangle;
    uint32_t local_id = 0;
    // triangle (triangle.hpp:12-19)
    V2 uv[3];
    V3 p[3], n[3], t[3], b[3];
    // analytic
    float radius = 0, length = 0;
    V3 center;
    M4 to_world;
};

struct Inst // instance.hpp:53-76 + blas.cpp:10-16
{
    uint32_t node_base = 0, prim_base = 0;
    uint32_t med_int = kNone, med_ext = kNone;
};

enum class Tex : uint32_t { kConstant = 1, kChecker, kBitmap };
struct Texture
{
    Tex type = Tex::kConstant;
    V3 color, color0, color1;
    M4 to_uv;
    int width = 0, height = 0, channel = 0;
    size_t texel_base = 0; // into Scene::texels
};

enum class Mat : uint32_t
{
    kAreaLight = 1, kDiffuse, kRoughDiffuse, kConductor, kDielectric,
    kThinDielectric, kPlastic
};
struct Bsdf
{
    Mat type = Mat::kDiffuse;
    bool twosided = false;
    uint32_t opacity = kNone, bump = kNone;
    uint32_t radiance = kNone;
    uint32_t reflectance = kNone, roughness = kNone;
    bool fast_approx = true;
    uint32_t rough_u = kNone, rough_v = kNone, spec_r = kNone, spec_t = kNone;
    V3 reflectivity3, edgetint, f_avg3;          // conductor
    float reflectivity = 1, eta = 1, eta_inv = 1; // dielectric / plastic
    float f_avg = 1, f_avg_inv = 1;
};

struct Medium // homogeneous.hpp:19-24 + medium.hpp:21-25
{
    float sampling_weight = 0;
    V3 sigma_s, sigma_t;
    bool hg = false;
    V3 g;
};

enum class Light : uint32_t { kPoint = 1, kSpot, kDirectional, kSun, kEnvMap, kConstant };
struct Emitter
{
    Light type = Light::kDirectional;
    V3 position, intensity, direction, radiance;
    float cutoff = 0, cos_cutoff = 0, uv_factor = 0, beam = 0, cos_beam = 0,
          transition_rcp = 0;
    uint32_t texture = kNone;
    M4 to_world, to_local;
    // env map: offsets into Scene::env_tables, pointed the way
    // emitter.cpp:166-175 points them (NOT the way renderer.cpp:597-606 wrote
    // the buffer — SURVEY.md quirk Q7).
    int width = 0, height = 0;
    float normalization = 0;
    size_t cdf_cols = 0, cdf_rows = 0, weight_rows = 0;
};

struct Camera // camera.cpp:26-37
{
    int width = 0, height = 0;
    uint32_t spp = 0;
    float spp_inv = 0;
    V3 eye, front, dx, dy;
};

constexpr int kLut = 128; // kulla_conty.hpp:9

struct Scene
{
    Camera camera;
    bool volpath = false, hide_emitters = false;
    float pdf_rr = 0.95f, rr_scale = 0.95f; // rr_scale: renderer.cpp:634 (Q2)
    uint32_t depth_rr = 5, depth_max = kNone;

    std::vector<Node> nodes;   // [TLAS | BLAS0 | BLAS1 | ...]
    std::vector<Prim> prims;
    std::vector<Inst> insts;
    std::vector<uint32_t> inst_bsdf;
    std::vector<float> inst_pdf_area;
    std::vector<uint32_t> light_inst;      // area light k -> instance
    std::vector<uint32_t> inst_light;      // instance -> area light or kNone
    std::vector<float> light_cdf;          // size n_lights + 1, NOT normalised

    std::vector<Texture> textures;
    std::vector<float> texels;
    std::vector<Bsdf> bsdfs;
    std::vector<Medium> media;
    std::vector<Emitter> emitters;
    std::vector<float> env_tables;
    uint32_t id_sun = kNone, id_envmap = kNone;

    std::vector<float> lut_brdf;   // kLut * kLut
    std::vector<float> lut_albedo; // kLut
};

// ---------------------------------------------------------------------------
// LBVH (bvh_builder.cpp).  Morton key = 30-bit code of the box centre
// relative to the union box, shifted left 32, OR-ed with the object index;
// objects sorted by key; recursive split at the highest differing key bit.
// Nodes are numbered in pre-order, children are tree-local indices.
// ---------------------------------------------------------------------------
namespace lbvh
{

inline uint32_t Spread10(uint32_t v) // bvh_builder.cpp:14-21
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

inline uint32_t Morton30(V3 v) // bvh_builder.cpp:38-48
{
    const float x = fminf(fmaxf(v.x * 1024.0f, 0.0f), 1023.0f),
                y = fminf(fmaxf(v.y * 1024.0f, 0.0f), 1023.0f),
                z = fminf(fmaxf(v.z * 1024.0f, 0.0f), 1023.0f);
    return Spread10(static_cast<uint32_t>(x)) * 4 +
           Spread10(static_cast<uint32_t>(y)) * 2 +
           Spread10(static_cast<uint32_t>(z));
}

inline int LeadingZeros64(uint64_t n) // bvh_builder.cpp:23-34
{
    int count = 0;
    for (int i = 63; i >= 0 && !((n >> i) & 1); --i)
        ++count;
    return count;
}

struct Builder
{
    const std::vector<Box> &boxes;
    const std::vector<float> &areas;
    std::vector<uint32_t> order;
    std::vector<uint64_t> keys;
    std::vector<Node> out;

    Builder(const std::vector<Box> &b, const std::vector<float> &a)
        : boxes(b), areas(a)
    {
        const uint32_t n = static_cast<uint32_t>(b.size());
        Box all;
        for (const Box &x : b)
            all.Grow(x);
        const V3 extent = all.hi - all.lo;
        keys.resize(n);
        order.resize(n);
        for (uint32_t i = 0; i < n; ++i)
        {
            const V3 rel = (b[i].Center() - all.lo) / extent;
            keys[i] = (static_cast<uint64_t>(Morton30(rel)) << 32) | i;
            order[i] = i;
        }
        std::sort(order.begin(), order.end(), [&](uint32_t p, uint32_t q)
                  { return keys[p] < keys[q]; });
        Emit(0, n);
    }

    uint32_t Split(uint32_t first, uint32_t last) const // bvh_builder.cpp:172-206
    {
        const uint64_t a = keys[order[first]], z = keys[order[last - 1]];
        if (a == z)
            return (first + last) >> 1;
        const int common = LeadingZeros64(a ^ z);
        uint32_t split = first, step = last - first;
        do
        {
            step = (step + 1) >> 1;
            const uint32_t probe = split + step;
            if (probe < last &&
                LeadingZeros64(a ^ keys[order[probe]]) > common)
                split = probe;
        } while (step > 1);
        return split;
    }

    uint32_t Emit(uint32_t begin, uint32_t end) // bvh_builder.cpp:143-170
    {
        const uint32_t id = static_cast<uint32_t>(out.size());
        if (begin + 1 > end)
            return kNone;
        if (begin + 1 == end)
        {
            Node leaf;
            leaf.leaf = true;
            leaf.object = order[begin];
            leaf.box = boxes[order[begin]];
            leaf.area = areas[order[begin]];
            out.push_back(leaf);
            return id;
        }
        Node inner;
        inner.leaf = false;
        out.push_back(inner);
        const uint32_t mid = Split(begin, end) + 1;
        const uint32_t l = Emit(begin, mid);
        const uint32_t r = Emit(mid, end);
        out[id].left = l;
        out[id].right = r;
        out[id].area = out[l].area + out[r].area;
        Box u;
        u.lo = Min3(out[l].box.lo, out[r].box.lo); // aabb.cpp:50-53
        u.hi = Max3(out[l].box.hi, out[r].box.hi);
        out[id].box = u;
        return id;
    }
};

inline std::vector<Node> Build(const std::vector<Box> &boxes,
                               const std::vector<float> &areas)
{
    return Builder(boxes, areas).out;
}

} // namespace lbvh

// ---------------------------------------------------------------------------
// Geometry commit
// ---------------------------------------------------------------------------
namespace commit
{

struct Mesh
{
    std::vector<V2> uv;
    std::vector<V3> p, n, t, b;
    std::vector<uint32_t> idx;
};

inline Mesh UnitRectangle() // scene.cpp:196-212
{
    Mesh m;
    m.uv = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    m.p = {{-1, -1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 1, 0}};
    m.n = {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}};
    m.idx = {0, 1, 2, 2, 3, 0};
    return m;
}

inline Mesh UnitCube() // scene.cpp:214-245: 6 faces x 4 vertices, 12 triangles
{
    Mesh m;
    static const float pos[24][3] = {
        {1, -1, -1}, {1, -1, 1},  {-1, -1, 1},  {-1, -1, -1}, {1, 1, -1},  {-1, 1, -1},
        {-1, 1, 1},  {1, 1, 1},   {1, -1, -1},  {1, 1, -1},   {1, 1, 1},   {1, -1, 1},
        {1, -1, 1},  {1, 1, 1},   {-1, 1, 1},   {-1, -1, 1},  {-1, -1, 1}, {-1, 1, 1},
        {-1, 1, -1}, {-1, -1, -1}, {1, 1, -1},  {1, -1, -1},  {-1, -1, -1}, {-1, 1, -1}};
    static const float face_normal[6][3] = {{0, -1, 0}, {0, 1, 0},  {1, 0, 0},
                                            {0, 0, 1},  {-1, 0, 0}, {0, 0, -1}};
    static const float corner_uv[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};
    for (int v = 0; v < 24; ++v)
    {
        m.p.push_back({pos[v][0], pos[v][1], pos[v][2]});
        const float *fn = face_normal[v / 4];
        m.n.push_back({fn[0], fn[1], fn[2]});
        m.uv.push_back({corner_uv[v % 4][0], corner_uv[v % 4][1]});
    }
    for (uint32_t f = 0; f < 6; ++f)
    {
        const uint32_t o = 4 * f;
        const uint32_t tri[6] = {o, o + 1, o + 2, o + 3, o, o + 2};
        m.idx.insert(m.idx.end(), tri, tri + 6);
    }
    return m;
}

inline Mesh FromMcsd(const mcsd::Instance &in)
{
    Mesh m;
    for (size_t k = 0; k + 1 < in.texcoords.size(); k += 2)
        m.uv.push_back({in.texcoords[k], in.texcoords[k + 1]});
    auto load3 = [](const std::vector<float> &src, std::vector<V3> *dst)
    {
        for (size_t k = 0; k + 2 < src.size(); k += 3)
            dst->push_back({src[k], src[k + 1], src[k + 2]});
    };
    load3(in.positions, &m.p);
    load3(in.normals, &m.n);
    load3(in.tangents, &m.t);
    load3(in.bitangents, &m.b);
    m.idx = in.indices;
    return m;
}

// scene.cpp:247-324 (bake to world space) + scene.cpp:15-111 (per-triangle
// records, |e1 x e2| as the "area" — quirk Q1 — and tangent frames).
inline void BakeMesh(Mesh m, const M4 &to_world, std::vector<Prim> *prims,
                     std::vector<float> *areas)
{
    if (m.idx.empty())
        throw std::runtime_error("mesh without indices");
    if (m.p.empty())
        throw std::runtime_error("mesh without positions");
    for (V3 &p : m.p)
        p = XformPoint(to_world, p);
    if (!m.n.empty())
    {
        const M4 normal_to_world = Inverse(Transpose(to_world));
        for (V3 &n : m.n)
            n = XformDir(normal_to_world, n);
    }
    for (V3 &t : m.t)
        t = XformDir(to_world, t);
    for (V3 &b : m.b)
        b = XformDir(to_world, b);

    const uint32_t count = static_cast<uint32_t>(m.idx.size() / 3);
    prims->assign(count, Prim());
    areas->assign(count, 0.0f);
    for (uint32_t i = 0; i < count; ++i)
    {
        Prim &tri = (*prims)[i];
        tri.shape = Shape::kTriangle;
        tri.local_id = i;
        const uint32_t *id = &m.idx[3 * i];
        if (m.uv.empty())
        {
            tri.uv[0] = {0, 0}, tri.uv[1] = {1, 0}, tri.uv[2] = {1, 1};
        }
        else
        {
            for (int j = 0; j < 3; ++j)
                tri.uv[j] = m.uv[id[j]];
        }
        for (int j = 0; j < 3; ++j)
            tri.p[j] = m.p[id[j]];
        const V3 e1 = tri.p[1] - tri.p[0], e2 = tri.p[2] - tri.p[0];
        const V3 ng = Cross(e1, e2);
        (*areas)[i] = Len(ng);
        if (m.n.empty())
        {
            const V3 flat = Unit(ng);
            for (int j = 0; j < 3; ++j)
                tri.n[j] = flat;
        }
        else
        {
            for (int j = 0; j < 3; ++j)
                tri.n[j] = m.n[id[j]];
        }
        if (m.t.empty() && m.b.empty())
        {
            const V2 d1 = tri.uv[1] - tri.uv[0], d2 = tri.uv[2] - tri.uv[0];
            const float r = 1.0f / (d1.v * d2.u - d1.u * d2.v);
            const V3 tangent = Unit((d1.v * e2 - d2.v * e1) * r);
            for (int j = 0; j < 3; ++j)
            {
                tri.b[j] = Unit(Cross(tri.n[j], tangent));
                tri.t[j] = Unit(Cross(tri.b[j], tri.n[j]));
            }
        }
        else if (m.t.empty())
        {
            for (int j = 0; j < 3; ++j)
            {
                tri.b[j] = m.b[id[j]];
                tri.t[j] = Unit(Cross(tri.b[j], tri.n[j]));
                tri.b[j] = Unit(Cross(tri.n[j], tri.t[j]));
            }
        }
        else
        {
            for (int j = 0; j < 3; ++j)
            {
                tri.t[j] = m.t[id[j]];
                tri.b[j] = Unit(Cross(tri.n[j], tri.t[j]));
                tri.t[j] = Unit(Cross(tri.b[j], tri.n[j]));
            }
        }
    }
}

inline Box PrimBox(const Prim &q)
{
    Box b;
    switch (q.shape)
    {
    case Shape::kTriangle: // triangle.cpp:9-15
        for (int j = 0; j < 3; ++j)
            b.Grow(q.p[j]);
        break;
    case Shape::kSphere: // sphere.cpp:9-15
        b.Grow(XformPoint(q.to_world, q.center + q.radius));
        b.Grow(XformPoint(q.to_world, q.center - q.radius));
        break;
    case Shape::kDisk: // disk.cpp:9-15
        b.Grow(XformPoint(q.to_world, V3{-0.5f, -0.5f, 0}));
        b.Grow(XformPoint(q.to_world, V3{0.5f, 0.5f, 0}));
        break;
    case Shape::kCylinder: // cylinder.cpp:9-19
        b.Grow(XformPoint(q.to_world, V3{q.radius, q.radius, 0}));
        b.Grow(XformPoint(q.to_world, V3{-q.radius, -q.radius, 0}));
        b.Grow(XformPoint(q.to_world, V3{q.radius, q.radius, q.length}));
        b.Grow(XformPoint(q.to_world, V3{-q.radius, -q.radius, q.length}));
        break;
    }
    return b;
}

} // namespace commit

// ---------------------------------------------------------------------------
// Kulla-Conty tables (kulla_conty.cpp:12-80).  Needs the isotropic GGX
// sampler and Smith G1, restated here because only the LUT build uses the
// isotropic sampler outside of plastic.
// ---------------------------------------------------------------------------
inline void GgxSampleIso(float xi0, float xi1, float alpha, V3 *h, float *pdf) // microfacet.cpp:8-19
{
    const float a2 = Sq(alpha);
    const float tan2 = a2 * xi0 / (1.0f - xi0), phi = k2Pi * xi1;
    const float cos_t = static_cast<float>(
        1.0f / sqrt(static_cast<double>(1.0f + tan2)));
    const float sin_t =
        static_cast<float>(sqrt(static_cast<double>(1.0f - Sq(cos_t))));
    *h = {sin_t * cosf(phi), sin_t * sinf(phi), cos_t};
    *pdf = static_cast<float>(
        1.0f / (static_cast<double>(kPi * a2) * pow(static_cast<double>(cos_t), 3) *
                static_cast<double>(Sq(1.0f + tan2 / a2))));
}

inline float SmithG1Iso(float alpha, V3 v, V3 h) // microfacet.cpp:62-74
{
    const float n_dot_v = v.z;
    if (n_dot_v * h.z <= 0)
        return 0;
    const float c2 = Sq(n_dot_v), tan2 = (1.0f - c2) / c2, a2 = Sq(alpha);
    return 2.0f / (1.0f + sqrtf(static_cast<float>(
                              1.0 + static_cast<double>(a2 * tan2))));
}

inline void BuildKullaConty(std::vector<float> *brdf, std::vector<float> *albedo)
{
    brdf->assign(kLut * kLut, 0.0f);
    albedo->assign(kLut, 0.0f);
    constexpr uint32_t kSamples = 1024;
    constexpr float kStepS = 1.0f / kSamples;
    const V3 n = {0.0f, 0.0f, 1.0f};
    const float step = 1.0f / kLut;
    // rows are independent; split them over threads (results unchanged)
    auto row = [&](int i)
    {
        float albedo_sum = 0.0f;
        const float alpha = step * (static_cast<float>(i) + 0.5f);
        for (int j = kLut - 1; j >= 0; --j)
        {
            const float mu = step * (static_cast<float>(j) + 0.5f);
            const V3 view = {-sqrtf(1.f - mu * mu), 0.0f, -mu};
            // IntegrateBRDF (kulla_conty.cpp:12-36)
            float acc = 0.0f;
            for (uint32_t s = 0; s < kSamples; ++s)
            {
                V3 h;
                float pdf_h;
                GgxSampleIso(s * kStepS, RadicalInverse<2>(s), alpha, &h, &pdf_h);
                const V3 l = Reflect(view, h);
                const float g = SmithG1Iso(alpha, -view, h) * SmithG1Iso(alpha, l, h);
                const float n_v = Dot(n, -view), n_l = Dot(n, l), n_h = Dot(n, h),
                            h_v = Dot(h, -view);
                if (n_l > 0.0f && n_h > 0.0f && h_v > 0.0f)
                    acc += (h_v * g) / (n_v * n_h);
            }
            const float e = fminf(acc * kStepS, 1.0f);
            (*brdf)[i * kLut + j] = e;
            // IntegrateAlbedo (kulla_conty.cpp:38-58)
            float acc2 = 0.0f;
            for (uint32_t s = 0; s < kSamples; ++s)
            {
                V3 h;
                float pdf_h;
                GgxSampleIso(s * kStepS, RadicalInverse<2>(s), alpha, &h, &pdf_h);
                const V3 l = Reflect(view, h);
                const float n_l = Dot(n, l), n_h = Dot(n, h), h_v = Dot(-view, h);
                if (n_l > 0.0f && n_h > 0.0f && h_v > 0.0f)
                    acc2 += e * n_l;
            }
            albedo_sum += acc2 * 2.0f * kStepS;
        }
        (*albedo)[i] = albedo_sum * step;
    };
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads < 1)
        n_threads = 1;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; ++t)
        pool.emplace_back([&, t]()
                          {
                              for (int i = kLut - 1 - static_cast<int>(t); i >= 0; i -= static_cast<int>(n_threads))
                                  row(i);
                          });
    for (std::thread &t : pool)
        t.join();
}

// bsdf.cpp:12-38: hemispherical average Fresnel of a dielectric interface.
inline float AvgFresnelDielectric(float eta)
{
    if (eta < 1.0)
        return -1.4399f * Sq(eta) + 0.7099f * eta + 0.6681f + 0.0636f / eta;
    const float i1 = 1.0f / eta, i2 = i1 * i1, i3 = i2 * i1, i4 = i3 * i1,
                i5 = i4 * i1;
    return 0.919317f - 3.4793f * i1 + 6.75335f * i2 - 7.80989f * i3 +
           4.98554f * i4 - 1.36881f * i5;
}

// bsdf.cpp:40-52: same for a conductor given reflectivity r and edge tint g.
inline V3 AvgFresnelConductor(V3 r, V3 g)
{
    return V3(0.087237f) + 0.0230685f * g - 0.0864902f * g * g +
           0.0774594f * g * g * g + 0.782654f * r - 0.136432f * r * r +
           0.278708f * r * r * r + 0.19744f * g * r + 0.0360605f * g * g * r -
           0.2586f * g * r * r;
}

V3 TextureColor(const Scene &sc, uint32_t id, V2 uv); // oracle_shading.hpp

inline float Luminance(V3 c) // envmap.cpp:9-12
{
    return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z;
}

// envmap.cpp:20-68; buffer written as [cdf_rows | weight_rows | cdf_cols]
// (renderer.cpp:597-606) and read back through the offsets of
// emitter.cpp:166-175 (cdf_cols -> +0, cdf_rows -> +(h+1), weight_rows ->
// +(2h+1)).
inline void BuildEnvTables(Scene *sc, Emitter *e, uint32_t id_tex)
{
    const Texture &tex = sc->textures[id_tex];
    if (tex.type != Tex::kBitmap)
        throw std::runtime_error("envmap radiance is not a bitmap");
    const int w = tex.width, h = tex.height;
    const float w_inv = 1.0f / w, h_inv = 1.0f / h;
    std::vector<float> cdf_rows(h + 1), weight_rows(h),
        cdf_cols(static_cast<size_t>(w + 1) * h);
    float sum_row = 0.0f;
    cdf_rows[0] = 0;
    for (int y = 0; y < h; ++y)
    {
        float sum_col = 0.0f;
        cdf_cols[0] = 0;
        for (int x = 0; x < w; ++x)
        {
            const V3 rgb = TextureColor(*sc, id_tex, {x * w_inv, y * h_inv});
            sum_col += Luminance(rgb);
            cdf_cols[static_cast<size_t>(y) * (w + 1) + (x + 1)] = sum_col;
        }
        cdf_cols[static_cast<size_t>(y) * (w + 1) + w] = 1.0f;
        const float norm_col = 1.0f / sum_col;
        for (int x = 1; x < w; ++x)
            cdf_cols[static_cast<size_t>(y) * (w + 1) + w - x] *= norm_col;
        const float weight = sinf((y + 0.5f) * kPi / h);
        weight_rows[y] = weight;
        sum_row += sum_col * weight;
        cdf_rows[y + 1] = sum_row;
    }
    cdf_rows[h] = 1.0f;
    const float norm_row = 1.0f / sum_row;
    for (int y = 1; y < h; ++y)
        cdf_rows[h - y] *= norm_row;
    if (!std::isfinite(sum_row))
        throw std::runtime_error("environment map contains nan/inf");
    const float normalization = static_cast<float>(
        1.0 / static_cast<double>(sum_row * (k2Pi * w_inv) * (kPi * h_inv)));

    const size_t base = sc->env_tables.size();
    sc->env_tables.insert(sc->env_tables.end(), cdf_rows.begin(), cdf_rows.end());
    sc->env_tables.insert(sc->env_tables.end(), weight_rows.begin(), weight_rows.end());
    sc->env_tables.insert(sc->env_tables.end(), cdf_cols.begin(), cdf_cols.end());
    e->width = w, e->height = h, e->normalization = normalization;
    e->cdf_cols = base;
    e->cdf_rows = base + static_cast<size_t>(h) + 1;
    e->weight_rows = base + static_cast<size_t>(h + 1) + h;
}

// ---------------------------------------------------------------------------
// Whole-scene commit: mcsd::Scene -> orc::Scene
// ---------------------------------------------------------------------------
inline void CommitScene(const mcsd::Scene &in, Scene *out)
{
    // The reference always builds the LUT (renderer.cpp:311-314, ~3.5 s);
    // only conductor and dielectric BSDFs read it, so skip it otherwise.
    bool build_lut = false;
    for (const mcsd::Bsdf &b : in.bsdfs)
        if (b.type == MCSD_BSDF_CONDUCTOR || b.type == MCSD_BSDF_DIELECTRIC)
            build_lut = true;
    Scene &sc = *out;
    sc = Scene();

    // camera (camera.cpp:26-37); fov_y is linear in angle (quirk Q4)
    {
        const mcsd::Camera &c = in.camera;
        Camera &cam = sc.camera;
        cam.width = c.width, cam.height = c.height, cam.spp = c.spp;
        cam.spp_inv = 1.0f / c.spp;
        const V3 eye = {c.eye[0], c.eye[1], c.eye[2]},
                 look_at = {c.look_at[0], c.look_at[1], c.look_at[2]},
                 up0 = {c.up[0], c.up[1], c.up[2]};
        const float fov_y = c.fov_x * c.height / c.width;
        cam.eye = eye;
        cam.front = Unit(look_at - eye);
        const V3 right = Unit(Cross(cam.front, up0));
        const V3 up = Unit(Cross(right, cam.front));
        cam.dx = right * tanf(Radians(0.5f * c.fov_x));
        cam.dy = up * tanf(Radians(0.5f * fov_y));
    }

    sc.volpath = in.integrator.type == MCSD_INTEGRATOR_VOLPATH;
    sc.hide_emitters = in.integrator.hide_emitters != 0;
    sc.pdf_rr = in.integrator.pdf_rr;
    sc.rr_scale = in.integrator.pdf_rr; // renderer.cpp:634: "rcp" holds pdf_rr itself
    sc.depth_rr = in.integrator.depth_rr;
    sc.depth_max = in.integrator.depth_max;

    // ---- geometry: one BLAS per instance, then the TLAS in front ----------
    std::vector<Node> blas_nodes;
    std::vector<uint32_t> node_base, prim_base;
    for (const mcsd::Instance &s : in.instances)
    {
        std::vector<Prim> prims;
        std::vector<float> areas;
        const M4 to_world = FromRowMajor(s.to_world);
        switch (s.type)
        {
        case MCSD_INST_RECTANGLE:
            commit::BakeMesh(commit::UnitRectangle(), to_world, &prims, &areas);
            break;
        case MCSD_INST_CUBE:
            commit::BakeMesh(commit::UnitCube(), to_world, &prims, &areas);
            break;
        case MCSD_INST_MESHES:
            commit::BakeMesh(commit::FromMcsd(s), to_world, &prims, &areas);
            break;
        case MCSD_INST_SPHERE: // scene.cpp:326-372
        {
            Prim q;
            q.shape = Shape::kSphere;
            q.radius = s.sphere_radius;
            q.center = {s.sphere_center[0], s.sphere_center[1], s.sphere_center[2]};
            q.to_world = to_world;
            prims = {q};
            const V3 cw = XformPoint(to_world, q.center),
                     bl = q.center + V3{q.radius, 0.0f, 0.0f},
                     bw = XformPoint(to_world, bl);
            const float rw = Len(cw - bw);
            areas = {4.0f * kPi * Sq(rw)};
            break;
        }
        case MCSD_INST_DISK: // scene.cpp:374-416
        {
            Prim q;
            q.shape = Shape::kDisk;
            q.to_world = to_world;
            prims = {q};
            const V3 cw = XformPoint(to_world, V3{0}),
                     bw = XformPoint(to_world, V3{0.5f, 0, 0});
            const float rw = Len(cw - bw);
            areas = {kPi * Sq(rw)};
            break;
        }
        case MCSD_INST_CYLINDER: // scene.cpp:418-472
        {
            Prim q;
            q.shape = Shape::kCylinder;
            const V3 p0 = {s.cyl_p0[0], s.cyl_p0[1], s.cyl_p0[2]},
                     p1 = {s.cyl_p1[0], s.cyl_p1[1], s.cyl_p1[2]};
            q.to_world = FrameMatrix(Unit(p1 - p0));
            q.to_world = MatMul(Translation(p0), q.to_world);
            q.to_world = MatMul(to_world, q.to_world);
            q.length = Len(XformPoint(q.to_world, {0, 0, Len(p1 - p0)}) -
                           XformPoint(q.to_world, {0, 0, 0}));
            q.radius = Len(XformPoint(q.to_world, {s.cyl_radius, 0, 0}) -
                           XformPoint(q.to_world, {0, 0, 0}));
            prims = {q};
            areas = {k2Pi * Sq(q.radius)};
            break;
        }
        default:
            throw std::runtime_error("unknown instance type");
        }
        std::vector<Box> boxes(prims.size());
        for (size_t i = 0; i < prims.size(); ++i)
            boxes[i] = commit::PrimBox(prims[i]);
        const std::vector<Node> tree = lbvh::Build(boxes, areas);
        node_base.push_back(static_cast<uint32_t>(blas_nodes.size()));
        prim_base.push_back(static_cast<uint32_t>(sc.prims.size()));
        blas_nodes.insert(blas_nodes.end(), tree.begin(), tree.end());
        sc.prims.insert(sc.prims.end(), prims.begin(), prims.end());
    }

    const uint32_t n_inst = static_cast<uint32_t>(in.instances.size());
    if (n_inst > 0) // scene.cpp:474-533
    {
        std::vector<Box> boxes(n_inst);
        std::vector<float> areas(n_inst);
        for (uint32_t i = 0; i < n_inst; ++i)
        {
            boxes[i] = blas_nodes[node_base[i]].box;
            areas[i] = blas_nodes[node_base[i]].area;
        }
        sc.inst_pdf_area.resize(n_inst);
        for (uint32_t i = 0; i < n_inst; ++i)
            sc.inst_pdf_area[i] = 1.0f / areas[i];
        const std::vector<Node> tlas = lbvh::Build(boxes, areas);
        sc.nodes = tlas;
        sc.nodes.insert(sc.nodes.end(), blas_nodes.begin(), blas_nodes.end());
        for (uint32_t i = 0; i < n_inst; ++i)
        {
            Inst it;
            it.node_base = node_base[i] + static_cast<uint32_t>(tlas.size());
            it.prim_base = prim_base[i];
            it.med_int = in.instances[i].id_medium_int;
            it.med_ext = in.instances[i].id_medium_ext;
            sc.insts.push_back(it);
        }
    }

    // ---- light tables (renderer.cpp:271-304) ------------------------------
    sc.inst_bsdf.resize(n_inst);
    sc.inst_light.assign(n_inst, kNone);
    std::vector<float> weights;
    for (uint32_t i = 0; i < n_inst; ++i)
    {
        const uint32_t b = in.instances[i].id_bsdf;
        sc.inst_bsdf[i] = b;
        if (b < in.bsdfs.size() && in.bsdfs[b].type == MCSD_BSDF_AREA_LIGHT)
        {
            sc.light_inst.push_back(i);
            weights.push_back(in.bsdfs[b].weight);
        }
    }
    sc.light_cdf.assign(weights.size() + 1, 0.0f);
    for (size_t k = 0; k < weights.size(); ++k)
    {
        sc.light_cdf[k + 1] = weights[k] + sc.light_cdf[k];
        sc.inst_light[sc.light_inst[k]] = static_cast<uint32_t>(k);
    }

    // ---- textures (renderer.cpp:371-431) ----------------------------------
    for (const mcsd::Texture &t : in.textures)
    {
        Texture o;
        switch (t.type)
        {
        case MCSD_TEX_CONSTANT:
            o.type = Tex::kConstant;
            o.color = {t.color[0], t.color[1], t.color[2]};
            break;
        case MCSD_TEX_CHECKERBOARD:
            o.type = Tex::kChecker;
            o.color0 = {t.color0[0], t.color0[1], t.color0[2]};
            o.color1 = {t.color1[0], t.color1[1], t.color1[2]};
            o.to_uv = FromRowMajor(t.to_uv);
            break;
        case MCSD_TEX_BITMAP:
            o.type = Tex::kBitmap;
            o.width = t.width, o.height = t.height, o.channel = t.channel;
            o.to_uv = FromRowMajor(t.to_uv);
            o.texel_base = sc.texels.size();
            sc.texels.insert(sc.texels.end(), t.data.begin(), t.data.end());
            break;
        default:
            throw std::runtime_error("unknown texture type");
        }
        sc.textures.push_back(o);
    }

    // ---- Kulla-Conty LUT ---------------------------------------------------
    if (build_lut)
        BuildKullaConty(&sc.lut_brdf, &sc.lut_albedo);
    else
    {
        sc.lut_brdf.assign(kLut * kLut, 0.0f);
        sc.lut_albedo.assign(kLut, 0.0f);
    }

    // ---- BSDF constants (bsdf.cpp:112-186) ---------------------------------
    const size_t n_tex = in.textures.size();
    auto check = [&](uint32_t id, bool allow_none)
    {
        if (id == kNone && allow_none)
            return;
        if (id >= n_tex)
            throw std::runtime_error("cannot find texture (id " +
                                     std::to_string(id) + ").");
    };
    for (const mcsd::Bsdf &b : in.bsdfs)
    {
        Bsdf o;
        o.twosided = b.twosided != 0;
        o.opacity = b.id_opacity, o.bump = b.id_bump_map;
        check(o.opacity, true), check(o.bump, true);
        switch (b.type)
        {
        case MCSD_BSDF_AREA_LIGHT:
            o.type = Mat::kAreaLight;
            o.radiance = b.id_radiance;
            check(o.radiance, false);
            break;
        case MCSD_BSDF_DIFFUSE:
            o.type = Mat::kDiffuse;
            o.reflectance = b.id_diffuse_reflectance;
            check(o.reflectance, false);
            break;
        case MCSD_BSDF_ROUGH_DIFFUSE:
            o.type = Mat::kRoughDiffuse;
            o.reflectance = b.id_diffuse_reflectance;
            o.roughness = b.id_roughness;
            // bsdf.cpp:139-144 never copies use_fast_approx into the committed
            // BSDF; the byte it would occupy is zero-initialised
            // (bsdf.cpp:66-70), so the full Oren-Nayar model always runs.
            o.fast_approx = false;
            check(o.reflectance, false), check(o.roughness, false);
            break;
        case MCSD_BSDF_CONDUCTOR:
            o.type = Mat::kConductor;
            o.rough_u = b.id_roughness_u, o.rough_v = b.id_roughness_v;
            o.spec_r = b.id_specular_reflectance;
            check(o.rough_u, false), check(o.rough_v, false), check(o.spec_r, false);
            o.reflectivity3 = {b.reflectivity[0], b.reflectivity[1], b.reflectivity[2]};
            o.edgetint = {b.edgetint[0], b.edgetint[1], b.edgetint[2]};
            o.f_avg3 = AvgFresnelConductor(o.reflectivity3, o.edgetint);
            break;
        case MCSD_BSDF_DIELECTRIC:
        case MCSD_BSDF_THIN_DIELECTRIC:
            o.type = b.type == MCSD_BSDF_DIELECTRIC ? Mat::kDielectric
                                                    : Mat::kThinDielectric;
            if (b.type == MCSD_BSDF_DIELECTRIC)
            {
                o.f_avg = AvgFresnelDielectric(b.eta);
                o.f_avg_inv = AvgFresnelDielectric(1.0f / b.eta);
            }
            o.twosided = true;
            o.rough_u = b.id_roughness_u, o.rough_v = b.id_roughness_v;
            o.spec_r = b.id_specular_reflectance;
            o.spec_t = b.id_specular_transmittance;
            check(o.rough_u, false), check(o.rough_v, false);
            check(o.spec_r, false), check(o.spec_t, false);
            o.eta = b.eta;
            o.eta_inv = 1.0f / b.eta;
            o.reflectivity = Sq(b.eta - 1.0f) / Sq(b.eta + 1.0f);
            break;
        case MCSD_BSDF_PLASTIC:
            o.type = Mat::kPlastic;
            o.roughness = b.id_roughness;
            o.reflectance = b.id_diffuse_reflectance;
            o.spec_r = b.id_specular_reflectance;
            check(o.roughness, false), check(o.reflectance, false), check(o.spec_r, false);
            o.reflectivity = Sq(b.eta - 1.0f) / Sq(b.eta + 1.0f);
            o.f_avg = AvgFresnelDielectric(b.eta);
            break;
        default:
            throw std::runtime_error("unknown BSDF type");
        }
        sc.bsdfs.push_back(o);
    }

    // ---- media (medium.cpp:6-39) ------------------------------------------
    for (const mcsd::Medium &m : in.media)
    {
        Medium o;
        const V3 sa = {m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]},
                 ss = {m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]};
        o.sigma_s = ss;
        o.sigma_t = sa + ss;
        const V3 albedo = ss / (sa + ss);
        for (int d = 0; d < 3; ++d)
            if (albedo[d] > o.sampling_weight && o.sigma_t[d] > 0)
                o.sampling_weight = albedo[d];
        if (o.sampling_weight > 0 && o.sampling_weight < 0.5f)
            o.sampling_weight = 0.5f;
        o.hg = m.phase_type == MCSD_PHASE_HG;
        o.g = {m.g[0], m.g[1], m.g[2]};
        sc.media.push_back(o);
    }

    // ---- emitters (emitter.cpp:122-175, renderer.cpp:520-620) --------------
    for (size_t i = 0; i < in.emitters.size(); ++i)
    {
        const mcsd::Emitter &e = in.emitters[i];
        Emitter o;
        switch (e.type)
        {
        case MCSD_EMIT_POINT:
            o.type = Light::kPoint;
            o.position = {e.position[0], e.position[1], e.position[2]};
            o.intensity = {e.intensity[0], e.intensity[1], e.intensity[2]};
            break;
        case MCSD_EMIT_SPOT:
            o.type = Light::kSpot;
            o.cutoff = e.cutoff_angle;
            o.cos_cutoff = cosf(e.cutoff_angle);
            o.uv_factor = tanf(e.cutoff_angle);
            o.beam = e.beam_width;
            o.cos_beam = cosf(e.beam_width);
            o.transition_rcp = 1.0f / (e.cutoff_angle - e.beam_width);
            o.intensity = {e.intensity[0], e.intensity[1], e.intensity[2]};
            o.texture = e.id_texture;
            o.to_world = FromRowMajor(e.to_world);
            o.position = XformPoint(o.to_world, {0, 0, 0});
            o.to_local = Inverse(o.to_world);
            break;
        case MCSD_EMIT_DIRECTIONAL:
            o.type = Light::kDirectional;
            o.direction = {e.direction[0], e.direction[1], e.direction[2]};
            o.radiance = {e.radiance[0], e.radiance[1], e.radiance[2]};
            break;
        case MCSD_EMIT_SUN:
            o.type = Light::kSun;
            o.cos_cutoff = e.cos_cutoff_angle;
            o.texture = e.id_texture;
            check(o.texture, false);
            o.direction = {e.direction[0], e.direction[1], e.direction[2]};
            o.radiance = {e.radiance[0], e.radiance[1], e.radiance[2]};
            sc.id_sun = static_cast<uint32_t>(i);
            break;
        case MCSD_EMIT_ENVMAP:
            o.type = Light::kEnvMap;
            o.texture = e.id_radiance;
            check(o.texture, false);
            o.to_world = FromRowMajor(e.to_world);
            o.to_local = Inverse(o.to_world);
            BuildEnvTables(&sc, &o, o.texture);
            sc.id_envmap = static_cast<uint32_t>(i);
            break;
        case MCSD_EMIT_CONSTANT:
            o.type = Light::kConstant;
            o.radiance = {e.radiance[0], e.radiance[1], e.radiance[2]};
            sc.id_envmap = static_cast<uint32_t>(i);
            break;
        default:
            throw std::runtime_error("unknown emitter type");
        }
        sc.emitters.push_back(o);
    }
}

} // namespace orc

#endif // ORACLE_SCENE_HPP
