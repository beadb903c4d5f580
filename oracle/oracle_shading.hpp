// TEST INFRASTRUCTURE — CPU oracle, part 3: textures, BSDFs, emitters, media.
//
// Function-for-function restatement of the reference's shading code with the
// random draws taken explicitly left to right (canonical clang order,
// SURVEY.md F5 and the draw ledger in §8 a-bis).  Conventions as in the
// reference: `wo` points from the surface towards the previous vertex, `wi`
// points INTO the surface (light travel direction); `attenuation` already
// contains f * cos(theta_i).
#ifndef ORACLE_SHADING_HPP
#define ORACLE_SHADING_HPP

#include "oracle_scene.hpp"

namespace orc
{

// ---------------------------------------------------------------------------
// Textures (texture.cpp:63-113, bitmap.cpp, checkboard.cpp, constant_texture.cpp)
// ---------------------------------------------------------------------------
inline void BitmapCoords(const Texture &t, V2 uv, uint32_t *x0, uint32_t *y0,
                         uint32_t *x1, uint32_t *y1, float *tx, float *ty)
{
    const V3 p = XformPoint(t.to_uv, {uv.u, uv.v, 0.0f});
    float x = p.x * t.width, y = p.y * t.height;
    while (x < 0)
        x += t.width;
    while (x > t.width - 1)
        x -= t.width;
    while (y < 0)
        y += t.height;
    while (y > t.height - 1)
        y -= t.height;
    *x0 = static_cast<uint32_t>(x), *y0 = static_cast<uint32_t>(y);
    *tx = x - *x0, *ty = y - *y0;
    *x1 = (*tx > 0.0f) ? *x0 + 1 : *x0;
    *y1 = (*ty > 0.0f) ? *y0 + 1 : *y0;
}

inline V3 BitmapColor(const Scene &sc, const Texture &t, V2 uv) // bitmap.cpp:6-55
{
    uint32_t x0, y0, x1, y1;
    float tx, ty;
    BitmapCoords(t, uv, &x0, &y0, &x1, &y1, &tx, &ty);
    const float *px = sc.texels.data() + t.texel_base;
    if (t.channel == 1)
    {
        const float c00 = px[x0 + t.width * y0], c01 = px[x0 + t.width * y1],
                    c10 = px[x1 + t.width * y0], c11 = px[x1 + t.width * y1];
        const float c0 = Mix(c00, c01, ty), c1 = Mix(c10, c11, ty);
        return V3(Mix(c0, c1, tx));
    }
    auto texel = [&](uint32_t x, uint32_t y)
    {
        const uint32_t o = (x + t.width * y) * t.channel;
        return V3{px[o], px[o + 1], px[o + 2]};
    };
    const V3 c00 = texel(x0, y0), c01 = texel(x0, y1), c10 = texel(x1, y0),
             c11 = texel(x1, y1);
    const V3 c0 = Mix(c00, c01, ty), c1 = Mix(c10, c11, ty);
    return Mix(c0, c1, tx);
}

inline V3 CheckerColor(const Texture &t, V2 uv) // checkboard.cpp:6-21
{
    V3 p = XformPoint(t.to_uv, {uv.u, uv.v, 0.0f});
    while (p.x > 1)
        p.x -= 1;
    while (p.x < 0)
        p.x += 1;
    while (p.y > 1)
        p.y -= 1;
    while (p.y < 0)
        p.y += 1;
    const int x = 2 * static_cast<int>(static_cast<int>(p.x * 2) % 2) - 1,
              y = 2 * static_cast<int>(static_cast<int>(p.y * 2) % 2) - 1;
    return (x * y == 1) ? t.color0 : t.color1;
}

inline V3 TextureColor(const Scene &sc, uint32_t id, V2 uv) // texture.cpp:63-78
{
    const Texture &t = sc.textures[id];
    switch (t.type)
    {
    case Tex::kConstant:
        return t.color;
    case Tex::kChecker:
        return CheckerColor(t, uv);
    case Tex::kBitmap:
        return BitmapColor(sc, t, uv);
    }
    return {};
}

inline V2 TextureGradient(const Scene &sc, uint32_t id, V2 uv) // texture.cpp:80-95
{
    const Texture &t = sc.textures[id];
    if (t.type == Tex::kConstant)
        return {};
    // bitmap.cpp:57-68 / checkboard.cpp:23-33: forward differences of |colour|
    constexpr float delta = 1e-4f, norm = 1.0f / delta;
    const float v = Len(TextureColor(sc, id, uv)),
                vu = Len(TextureColor(sc, id, uv + V2{delta, 0})),
                vv = Len(TextureColor(sc, id, uv + V2{0, delta}));
    return {(vu - v) * norm, (vv - v) * norm};
}

// texture.cpp:97-113: stochastic opacity.  Draws one random number for a
// constant texture and for a 4-channel bitmap, none otherwise.
inline bool TextureTransparent(const Scene &sc, uint32_t id, V2 uv, uint32_t *rng)
{
    const Texture &t = sc.textures[id];
    switch (t.type)
    {
    case Tex::kConstant:
        return t.color.x < Rand(rng);
    case Tex::kChecker:
        return false;
    case Tex::kBitmap:
    {
        if (t.channel != 4)
            return false;
        uint32_t x0, y0, x1, y1;
        float tx, ty;
        BitmapCoords(t, uv, &x0, &y0, &x1, &y1, &tx, &ty);
        const float *px = sc.texels.data() + t.texel_base;
        const float c00 = px[(x0 + t.width * y0) * 4 + 3],
                    c01 = px[(x0 + t.width * y1) * 4 + 3],
                    c10 = px[(x1 + t.width * y0) * 4 + 3],
                    c11 = px[(x1 + t.width * y1) * 4 + 3];
        const float c0 = Mix(c00, c01, ty), c1 = Mix(c10, c11, ty);
        return Mix(c0, c1, tx) < Rand(rng);
    }
    }
    return false;
}

// ---------------------------------------------------------------------------
// BSDF plumbing (bsdf.hpp:84-100, bsdf.cpp:100-110,238-276)
// ---------------------------------------------------------------------------
struct Scatter
{
    bool valid = false, inside = false;
    float pdf = 0;
    V2 uv;
    V3 wi, wo, position, normal, tangent, bitangent, attenuation;

    V3 Local(V3 v) const { return Unit({Dot(v, tangent), Dot(v, bitangent), Dot(v, normal)}); }
    V3 World(V3 v) const { return Unit(v.x * tangent + v.y * bitangent + v.z * normal); }
};

inline bool BsdfTransparent(const Scene &sc, const Bsdf &b, V2 uv, uint32_t *rng)
{
    return b.opacity != kNone && TextureTransparent(sc, b.opacity, uv, rng);
}

inline V3 BsdfBump(const Scene &sc, const Bsdf &b, V3 n, V3 t, V3 bt, V2 uv)
{
    if (b.bump == kNone)
        return n;
    const V2 g = TextureGradient(sc, b.bump, uv);
    return Unit(-g.u * t - g.v * bt + n);
}

inline V3 BsdfRadiance(const Scene &sc, const Bsdf &b, V2 uv)
{
    if (b.type == Mat::kAreaLight)
        return TextureColor(sc, b.radiance, uv);
    return {};
}

// ---------------------------------------------------------------------------
// GGX helpers (microfacet.cpp, microfacet.hpp:24-29)
// ---------------------------------------------------------------------------
inline void GgxSampleAniso(float xi0, float xi1, float au, float av, V3 *h,
                           float *pdf) // microfacet.cpp:21-35
{
    const float phi = (atanf(av / au * tanf(kPi + k2Pi * xi1)) +
                       kPi * floorf(2.0f * xi1 + 0.5f));
    const float cos_p = cosf(phi), sin_p = sinf(phi),
                a2 = 1.0f / (Sq(cos_p / au) + Sq(sin_p / av));
    const float tan2 = static_cast<float>(static_cast<double>(a2 * xi0) /
                                          (1.0 - static_cast<double>(xi0)));
    const float cos_t = 1.0f / sqrtf(1.0f + tan2),
                sin_t = sqrtf(1.0f - Sq(cos_t));
    *h = {sin_t * cos_p, sin_t * sin_p, cos_t};
    *pdf = static_cast<float>(
        1.0f / (static_cast<double>(kPi * au * av) *
                pow(static_cast<double>(cos_t), 3) *
                static_cast<double>(Sq(1.0f + tan2 / a2))));
}

inline float GgxPdfIso(float alpha, V3 h) // microfacet.cpp:37-46
{
    const float c = h.z;
    if (c <= 0.0f)
        return 0.0f;
    const float c2 = Sq(c), tan2 = (1.0f - c2) / c2;
    const float c3 = static_cast<float>(pow(static_cast<double>(c), 3));
    const float a2 = Sq(alpha);
    return a2 / (kPi * c3 * Sq(a2 + tan2));
}

inline float GgxPdfAniso(float au, float av, V3 h) // microfacet.cpp:48-59
{
    const float c = h.z;
    if (c <= 0.0f)
        return 0.0f;
    const float c2 = Sq(c);
    return c / (kPi * au * av * Sq(Sq(h.x / au) + Sq(h.y / av) + c2));
}

inline float SmithG1Aniso(float au, float av, V3 v, V3 h) // microfacet.cpp:76-86
{
    const float n_dot_v = v.z;
    if (n_dot_v * h.z <= 0)
        return 0;
    const float xy = Sq(au * v.x) + Sq(av * v.y), tan2 = xy / Sq(n_dot_v);
    return 2.0f / (1.0f + sqrtf(1.0f + tan2));
}

// microfacet.hpp:24-29: (1 - r) * float(pow(1 - cos, 5)) + r, pow in double.
inline float Pow5(float cos_t)
{
    return static_cast<float>(pow(static_cast<double>(1.0f - cos_t), 5));
}
inline float Schlick(float cos_t, float r) { return (1.0f - r) * Pow5(cos_t) + r; }
inline V3 Schlick(float cos_t, V3 r) { return (1.0f - r) * Pow5(cos_t) + r; }

// ---------------------------------------------------------------------------
// Kulla-Conty lookups (kulla_conty.cpp:82-143)
// ---------------------------------------------------------------------------
// kLutResolution is an UNSIGNED constant in the reference
// (kulla_conty.hpp:9), so the range checks `offset_int >= kLutResolution - 1`
// compare as unsigned: a NEGATIVE index (dielectric.cpp:211-212 passes
// N_dot_O < 0 for transmitted connections) wraps to a huge value and takes
// the "last column / last row" branch instead of reading out of bounds.
inline float LutBrdf(const Scene &sc, float cos_t, float alpha)
{
    const float *t = sc.lut_brdf.data();
    const float o1 = alpha * kLut, o2 = cos_t * kLut;
    const int i1 = static_cast<int>(o1), i2 = static_cast<int>(o2);
    constexpr uint32_t last = kLut - 1;
    const bool row_clamped = static_cast<uint32_t>(i1) >= last,
               col_clamped = static_cast<uint32_t>(i2) >= last;
    if (row_clamped)
    {
        if (col_clamped)
            return t[last * kLut + last];
        return Mix(t[last * kLut + i2], t[last * kLut + i2 + 1], o2 - i2);
    }
    if (col_clamped)
        return Mix(t[i1 * kLut + last], t[(i1 + 1) * kLut + last], o1 - i1);
    return Mix(Mix(t[i1 * kLut + i2], t[(i1 + 1) * kLut + i2], o1 - i1),
               Mix(t[i1 * kLut + i2 + 1], t[(i1 + 1) * kLut + i2 + 1], o1 - i1),
               o2 - i2);
}

inline float LutAlbedo(const Scene &sc, float alpha)
{
    const float *t = sc.lut_albedo.data();
    const float o = alpha * kLut;
    const int i = static_cast<int>(o);
    if (static_cast<uint32_t>(i) >= static_cast<uint32_t>(kLut - 1)) // unsigned compare, see LutBrdf
        return t[kLut - 1];
    return Mix(t[i], t[i + 1], o - i);
}

// ---------------------------------------------------------------------------
// Diffuse (diffuse.cpp)
// ---------------------------------------------------------------------------
inline void DiffuseEval(const Scene &sc, const Bsdf &b, Scatter *r) // diffuse.cpp:9-20
{
    r->pdf = Dot(r->wo, r->normal); // outgoing cosine, no 1/pi (quirk Q11)
    if (r->pdf < kEps)
        return;
    r->valid = true;
    const V3 albedo = TextureColor(sc, b.reflectance, r->uv);
    const float n_i = Dot(-r->wi, r->normal);
    r->attenuation = albedo * k1DivPi * n_i;
}

inline void DiffuseSample(const Scene &sc, const Bsdf &b, uint32_t *rng, Scatter *r) // diffuse.cpp:22-34
{
    const float xi0 = Rand(rng), xi1 = Rand(rng);
    V3 local;
    HemisphereCosine(xi0, xi1, &local, &r->pdf);
    if (r->pdf < kEps)
        return;
    r->wi = -r->World(local);
    r->valid = true;
    const V3 albedo = TextureColor(sc, b.reflectance, r->uv);
    r->attenuation = albedo * k1DivPi * local.z;
}

// ---------------------------------------------------------------------------
// Oren-Nayar rough diffuse (rough_diffuse.cpp)
// ---------------------------------------------------------------------------
inline void OrenNayar(float rough, V3 albedo, bool fast, Scatter *r) // rough_diffuse.cpp:10-95
{
    constexpr float conv = 0.70710678118f;
    const float s2 = Sq(rough * conv);
    const V3 li = r->Local(-r->wi), lo = r->Local(r->wo);
    const float n_i = li.z, n_o = lo.z, sin_i = sqrtf(1.0f - n_i * n_i),
                sin_o = sqrtf(1.0f - n_o * n_o);
    float phi_i, theta_i, phi_o, theta_o;
    ToSpherical(li, &theta_i, &phi_i, nullptr);
    ToSpherical(lo, &theta_o, &phi_o, nullptr);
    const float cos_dphi = cosf(phi_i) * cosf(phi_o) + sinf(phi_i) * sinf(phi_o);
    if (fast)
    {
        const float A = 1.0f - 0.5f * s2 / (s2 + 0.33f),
                    B = 0.45f * s2 / (s2 + 0.09f);
        float sin_a, tan_b;
        if (n_i > n_o)
            sin_a = sin_o, tan_b = sin_i / n_i;
        else
            sin_a = sin_i, tan_b = sin_o / n_o;
        r->attenuation = albedo * k1DivPi * n_i *
                         (A + B * fmaxf(cos_dphi, 0.0f) * sin_a * tan_b);
        return;
    }
    const float alpha = fmaxf(theta_i, theta_o), beta = fminf(theta_i, theta_o);
    float sin_a, sin_b, tan_b;
    if (n_i > n_o)
        sin_a = sin_o, sin_b = sin_i, tan_b = sin_i / n_i;
    else
        sin_a = sin_i, sin_b = sin_o, tan_b = sin_o / n_o;
    const float tmp = s2 / (s2 + 0.09f),
                tmp2 = 4.0f * k1DivPi * k1DivPi * alpha * beta,
                tmp3 = 2.0f * beta * k1DivPi;
    const float C1 = 1.0f - 0.5f * s2 / (s2 + 0.33f);
    float C2 = 0.45f * tmp;
    const float C3 = 0.125f * tmp * tmp2 * tmp2, C4 = 0.17f * s2 / (s2 + 0.13f);
    if (cos_dphi > 0)
        C2 *= sin_a;
    else // float *= (float - double pow) : product formed in double
        C2 = static_cast<float>(static_cast<double>(C2) *
                                (static_cast<double>(sin_a) -
                                 pow(static_cast<double>(tmp3), 3)));
    // (float + float) / (float + double sqrt) -> double -> float
    const float tan_half = static_cast<float>(
        static_cast<double>(sin_a + sin_b) /
        (static_cast<double>(sqrtf(fmaxf(0.0f, 1.0f - Sq(sin_a)))) +
         sqrt(static_cast<double>(fmaxf(0.0f, 1.0f - Sq(sin_b))))));
    // C1 + cos*C2*tan_b (float) + (1 - |cos|) * C3 * tan_half (double)
    const double single_d =
        static_cast<double>(C1 + cos_dphi * C2 * tan_b) +
        (1.0f - fabs(static_cast<double>(cos_dphi))) * C3 * tan_half;
    const V3 single = albedo * static_cast<float>(single_d),
             dbl = Sq(albedo) * (C4 * (1.0f - cos_dphi * Sq(tmp3)));
    r->attenuation = (single + dbl) * k1DivPi * n_i;
}

inline void RoughDiffuseEval(const Scene &sc, const Bsdf &b, Scatter *r) // rough_diffuse.cpp:117-128
{
    r->pdf = Dot(r->wo, r->normal);
    if (r->pdf < kEps)
        return;
    r->valid = true;
    const float alpha = TextureColor(sc, b.roughness, r->uv).x;
    const V3 albedo = TextureColor(sc, b.reflectance, r->uv);
    OrenNayar(alpha, albedo, b.fast_approx, r);
}

inline void RoughDiffuseSample(const Scene &sc, const Bsdf &b, uint32_t *rng, Scatter *r) // rough_diffuse.cpp:99-115
{
    const float xi0 = Rand(rng), xi1 = Rand(rng);
    V3 local;
    HemisphereCosine(xi0, xi1, &local, &r->pdf);
    if (r->pdf < kEps)
        return;
    r->wi = -Unit(local.x * r->tangent + local.y * r->bitangent + local.z * r->normal);
    r->valid = true;
    const float alpha = TextureColor(sc, b.roughness, r->uv).x;
    const V3 albedo = TextureColor(sc, b.reflectance, r->uv);
    OrenNayar(alpha, albedo, b.fast_approx, r);
}

// ---------------------------------------------------------------------------
// Conductor (conductor.cpp)
// ---------------------------------------------------------------------------
inline V3 ConductorMs(const Scene &sc, const Bsdf &b, float n_i, float n_o, float alpha) // conductor.cpp:14-28
{
    const float e_i = LutBrdf(sc, n_i, alpha), e_o = LutBrdf(sc, n_o, alpha),
                e_avg = LutAlbedo(sc, alpha),
                f_ms = (1.0f - e_i) * (1.0f - e_o) / (kPi * (1.0f - e_avg));
    const V3 f_add = Sq(b.f_avg3) * e_avg / (1.0f - b.f_avg3 * (1.0f - e_avg));
    return f_ms * f_add * n_i;
}

inline void ConductorSample(const Scene &sc, const Bsdf &b, uint32_t *rng, Scatter *r) // conductor.cpp:34-77
{
    const float au = TextureColor(sc, b.rough_u, r->uv).x,
                av = TextureColor(sc, b.rough_v, r->uv).x;
    const float xi0 = Rand(rng), xi1 = Rand(rng);
    V3 h_local(0);
    float D = 0;
    GgxSampleAniso(xi0, xi1, au, av, &h_local, &D);
    const V3 h = r->World(h_local);
    const float h_o = Dot(r->wo, h);
    r->pdf = D / (4.0f * h_o);
    if (r->pdf < kEps)
        return;
    r->wi = -Reflect(-r->wo, h);
    const float n_i = Dot(-r->wi, r->normal);
    if (n_i < kEpsFloat)
        return;
    r->valid = true;
    const V3 li = r->Local(-r->wi), lo = r->Local(r->wo);
    const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local),
                h_i = Dot(-r->wi, h), n_o = lo.z;
    const V3 F = Schlick(h_i, b.reflectivity3);
    r->attenuation = (F * D * G) / (4.0f * n_o);
    if (au == av)
        r->attenuation += ConductorMs(sc, b, n_i, n_o, au);
    r->attenuation *= TextureColor(sc, b.spec_r, r->uv);
}

inline void ConductorEval(const Scene &sc, const Bsdf &b, Scatter *r) // conductor.cpp:79-119
{
    const float n_o = Dot(r->wo, r->normal);
    if (n_o < kEpsFloat)
        return;
    const V3 h = Unit(-r->wi + r->wo), h_local = r->Local(h);
    const float au = TextureColor(sc, b.rough_u, r->uv).x,
                av = TextureColor(sc, b.rough_v, r->uv).x,
                D = GgxPdfAniso(au, av, h_local), h_o = Dot(r->wo, h);
    r->pdf = D / (4.0f * h_o);
    if (r->pdf < kEps)
        return;
    r->valid = true;
    const V3 li = r->Local(-r->wi), lo = r->Local(r->wo);
    const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local),
                h_i = Dot(-r->wi, h);
    const V3 F = Schlick(h_i, b.reflectivity3);
    r->attenuation = (F * D * G) / (4.0f * n_o);
    if (au == av)
    {
        const float n_i = Dot(-r->wi, r->normal);
        r->attenuation += ConductorMs(sc, b, n_i, n_o, au);
    }
    r->attenuation *= TextureColor(sc, b.spec_r, r->uv);
}

// ---------------------------------------------------------------------------
// Dielectric (dielectric.cpp)
// ---------------------------------------------------------------------------
inline float DielectricMs(const Scene &sc, const Bsdf &b, float n_i, float n_o,
                          float alpha, bool inside, bool reflect) // dielectric.cpp:14-38
{
    const float e_i = LutBrdf(sc, n_i, alpha), e_o = LutBrdf(sc, n_o, alpha),
                e_avg = LutAlbedo(sc, alpha),
                f_ms = (1.0f - e_i) * (1.0f - e_o) / (kPi * (1.0f - e_avg));
    const float F = inside ? b.f_avg_inv : b.f_avg, eta = inside ? b.eta_inv : b.eta;
    // pow(float, 2) is the double pow; the surrounding products are double.
    const float f_add = static_cast<float>(
        pow(static_cast<double>(F), 2) * e_avg /
        static_cast<double>(1.0f - F * (1.0f - e_avg)));
    const double eta2 = pow(static_cast<double>(eta), 2);
    const float ratio_t = static_cast<float>(
        (static_cast<double>((1.0f - b.f_avg) * (1.0f - b.f_avg_inv)) * eta2 /
         (static_cast<double>(1.0f - b.f_avg) +
          static_cast<double>(1.0f - b.f_avg_inv) * eta2)));
    const float ret = f_ms * f_add * n_i;
    return reflect ? (1.0f - ratio_t) * ret : ratio_t * ret;
}

inline void DielectricSample(const Scene &sc, const Bsdf &b, uint32_t *rng, Scatter *r) // dielectric.cpp:44-140
{
    // sampling roughness is scaled, evaluation roughness is not (quirk Q10);
    // sqrt is the double one, abs resolves to the float overload (F4).
    const float scale = static_cast<float>(
        1.2f - 0.2f * sqrt(static_cast<double>(fabsf(Dot(-r->wo, r->normal)))));
    const float au = TextureColor(sc, b.rough_u, r->uv).x * scale,
                av = TextureColor(sc, b.rough_v, r->uv).x * scale;
    const float xi0 = Rand(rng), xi1 = Rand(rng);
    V3 h_local(0);
    float D = 0;
    GgxSampleAniso(xi0, xi1, au, av, &h_local, &D);
    const V3 h = r->World(h_local);
    float h_o = Dot(r->wo, h);
    if (h_o < kEpsFloat)
        return;
    float eta = b.eta, eta_inv = b.eta_inv;
    if (!r->inside)
    {
        const float t = eta_inv;
        eta_inv = eta;
        eta = t;
    }
    V3 wt;
    const bool total_reflection = !Refract(-r->wo, h, eta, &wt);
    float F = Schlick(h_o, b.reflectivity);
    const V3 lo = r->Local(r->wo);
    if (total_reflection || Rand(rng) < F)
    {
        r->wi = -Reflect(-r->wo, h);
        const float n_i = Dot(-r->wi, r->normal);
        if (n_i < kEpsFloat)
            return;
        r->pdf = F * D / (4.0f * h_o);
        if (r->pdf < kEps)
            return;
        const V3 li = r->Local(-r->wi);
        const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local),
                    n_o = lo.z;
        r->attenuation = V3((F * D * G) / (4.0f * n_o));
        if (au == av)
            r->attenuation += V3(DielectricMs(sc, b, n_i, n_o, au, r->inside, true));
        r->attenuation *= TextureColor(sc, b.spec_r, r->uv);
    }
    else
    {
        r->wi = -wt;
        V3 li = r->Local(-r->wi);
        li.z = -li.z;
        const float n_i = li.z;
        if (n_i < kEpsFloat)
            return;
        const float h_i = -Dot(wt, h);
        if (h_i < kEpsFloat)
            return;
        h_o = -h_o;
        F = Schlick(h_i, b.reflectivity);
        r->pdf = ((1.0f - F) * D) * fabsf(h_o / Sq(eta_inv * h_i + h_o));
        if (r->pdf < kEps)
            return;
        const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local),
                    n_o = lo.z;
        r->attenuation = V3(((fabsf(h_i) * fabsf(h_o)) * ((1.0f - F) * G * D)) /
                            fabsf(n_o * Sq(eta_inv * h_i + h_o)));
        if (au == av)
            r->attenuation += V3(DielectricMs(sc, b, n_i, n_o, au, !r->inside, false));
        r->attenuation *= Sq(eta);
        r->attenuation *= TextureColor(sc, b.spec_t, r->uv);
    }
    r->valid = true;
}

inline void DielectricEval(const Scene &sc, const Bsdf &b, Scatter *r) // dielectric.cpp:142-224
{
    float eta = b.eta, eta_inv = b.eta_inv;
    if (r->inside)
    {
        const float t = eta_inv;
        eta_inv = eta;
        eta = t;
    }
    const float n_o = Dot(r->wo, r->normal);
    const bool reflect = n_o > 0.0f;
    const V3 h = reflect ? Unit(-r->wi + r->wo) : -Unit(eta_inv * (-r->wi) + r->wo),
             h_local = r->Local(h);
    const float au = TextureColor(sc, b.rough_u, r->uv).x,
                av = TextureColor(sc, b.rough_v, r->uv).x,
                D = GgxPdfAniso(au, av, h_local), h_i = Dot(-r->wi, h),
                h_o = Dot(r->wo, h), F = Schlick(h_i, b.reflectivity);
    r->pdf = reflect ? (F * D) / (4.0f * h_o)
                     : (((1.0f - F) * D) * fabsf(h_o / Sq(eta_inv * h_i + h_o)));
    if (r->pdf < kEps)
        return;
    r->valid = true;
    const V3 li = r->Local(-r->wi);
    if (reflect)
    {
        const V3 lo = r->Local(r->wo);
        const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local);
        r->attenuation = V3((F * D * G) / (4.0f * n_o));
        if (au == av)
        {
            const float n_i = Dot(-r->wi, r->normal);
            r->attenuation += V3(DielectricMs(sc, b, n_i, n_o, au, r->inside, true));
        }
        r->attenuation *= TextureColor(sc, b.spec_r, r->uv);
    }
    else
    {
        const V3 lo = r->Local(-r->wo);
        const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local);
        r->attenuation = V3(((fabsf(h_i) * fabsf(h_o)) * ((1.0f - F) * G * D)) /
                            fabsf(n_o * Sq(eta_inv * h_i + h_o)));
        if (au == av)
        {
            const float n_i = Dot(r->normal, -r->wi);
            r->attenuation += V3(DielectricMs(sc, b, n_i, n_o, au, r->inside, false));
        }
        r->attenuation *= Sq(eta);
        r->attenuation *= TextureColor(sc, b.spec_t, r->uv);
    }
}

// ---------------------------------------------------------------------------
// Thin dielectric (thin_dielectric.cpp)
// ---------------------------------------------------------------------------
inline void ThinDielectricSample(const Scene &sc, const Bsdf &b, uint32_t *rng, Scatter *r) // :11-67
{
    const float au = TextureColor(sc, b.rough_u, r->uv).x,
                av = TextureColor(sc, b.rough_v, r->uv).x;
    const float xi0 = Rand(rng), xi1 = Rand(rng);
    V3 h_local(0);
    float D = 0;
    GgxSampleAniso(xi0, xi1, au, av, &h_local, &D);
    const V3 h = r->World(h_local);
    const float h_o = Dot(r->wo, h);
    r->pdf = D / (4.0f * h_o);
    if (r->pdf < kEps)
        return;
    r->wi = -Reflect(-r->wo, h);
    const float n_i = Dot(-r->wi, r->normal);
    if (n_i < kEpsFloat)
        return;
    const V3 li = r->Local(-r->wi), lo = r->Local(r->wo);
    const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local),
                h_i = Dot(-r->wi, h), n_o = lo.z;
    float F = Schlick(h_i, b.reflectivity);
    if (F < 1.0f)
        F *= 2.0f / (1.0f + F);
    if (Rand(rng) < F)
    {
        r->pdf *= F;
        if (r->pdf < kEps)
            return;
        r->attenuation = V3((F * D * G) / (4.0f * n_o));
        r->attenuation *= TextureColor(sc, b.spec_r, r->uv);
    }
    else
    {
        r->pdf *= 1.0f - F;
        if (r->pdf < kEps)
            return;
        r->attenuation = V3(((1.0f - F) * D * G) / (4.0f * n_o));
        r->attenuation *= TextureColor(sc, b.spec_t, r->uv);
        r->wi = r->wo;
    }
    r->valid = true;
}

inline void ThinDielectricEval(const Scene &sc, const Bsdf &b, Scatter *r) // :69-124
{
    bool reflect = true;
    V3 wo = r->wo;
    float n_o = Dot(r->wo, r->normal);
    if (fabs(static_cast<double>(n_o)) < kEpsFloat)
        return;
    V3 lo = r->Local(r->wo);
    if (n_o < 0.0f)
    {
        reflect = false;
        n_o = -n_o;
        lo.z = -lo.z;
        wo = r->World(lo);
    }
    const V3 h = Unit(-r->wi + wo), h_local = r->Local(h);
    const float au = TextureColor(sc, b.rough_u, r->uv).x,
                av = TextureColor(sc, b.rough_v, r->uv).x,
                D = GgxPdfAniso(au, av, h_local), h_i = Dot(-r->wi, h),
                h_o = Dot(r->wo, h);
    float F = Schlick(h_i, b.reflectivity);
    if (F < 1.0f)
        F *= 2.0f / (1.0f + F);
    r->pdf = reflect ? (F * D) / (4.0f * h_o) : ((1.0f - F) * D) / (4.0f * h_o);
    if (r->pdf < kEps)
        return;
    r->valid = true;
    const V3 li = r->Local(-r->wi);
    const float G = SmithG1Aniso(au, av, li, h_local) * SmithG1Aniso(au, av, lo, h_local);
    if (reflect)
    {
        r->attenuation = V3((F * D * G) / (4.0f * n_o));
        r->attenuation *= TextureColor(sc, b.spec_r, r->uv);
    }
    else
    {
        r->attenuation = V3(((1.0f - F) * D * G) / (4.0f * n_o));
        r->attenuation *= TextureColor(sc, b.spec_t, r->uv);
    }
}

// ---------------------------------------------------------------------------
// Plastic (plastic.cpp)
// ---------------------------------------------------------------------------
inline void PlasticSample(const Scene &sc, const Bsdf &b, uint32_t *rng, Scatter *r) // plastic.cpp:11-96
{
    const V3 kd = TextureColor(sc, b.reflectance, r->uv),
             ks = TextureColor(sc, b.spec_r, r->uv);
    const float w_spec = (ks.x + ks.y + ks.z) / ((kd.x + kd.y + kd.z) + (ks.x + ks.y + ks.z));
    const float n_o = Dot(r->wo, r->normal), kr_o = Schlick(n_o, b.reflectivity);
    float kr_i = kr_o, pdf_spec = kr_i * w_spec,
          pdf_diff = (1.0f - kr_i) * (1.0f - w_spec);
    pdf_spec = pdf_spec / (pdf_spec + pdf_diff);
    pdf_diff = 1.0f - pdf_spec;
    V3 h_local(0), h(0);
    float D = 0;
    const float alpha = TextureColor(sc, b.roughness, r->uv).x;
    float n_i = 0;
    if (Rand(rng) < pdf_spec)
    {
        const float xi0 = Rand(rng), xi1 = Rand(rng);
        GgxSampleIso(xi0, xi1, alpha, &h_local, &D);
        h = r->World(h_local);
        r->wi = -Reflect(-r->wo, h);
        n_i = Dot(-r->wi, r->normal);
        if (n_i < kEpsFloat)
            return;
        kr_i = Schlick(n_i, b.reflectivity);
        pdf_spec = kr_i * w_spec, pdf_diff = (1.0f - kr_i) * w_spec;
        pdf_spec = pdf_spec / (pdf_spec + pdf_diff), pdf_diff = 1.0f - pdf_spec;
        const float h_o = Dot(r->wo, h);
        pdf_spec *= D / (4.0f * h_o);
        pdf_diff *= Dot(-r->wi, r->normal);
    }
    else
    {
        const float xi0 = Rand(rng), xi1 = Rand(rng);
        V3 local(0);
        float pdf_local = 0.0f;
        HemisphereCosine(xi0, xi1, &local, &pdf_local);
        r->wi = -r->World(local);
        n_i = Dot(-r->wi, r->normal);
        kr_i = Schlick(n_i, b.reflectivity);
        pdf_spec = kr_i * w_spec, pdf_diff = (1.0f - kr_i) * w_spec;
        pdf_spec = pdf_spec / (pdf_spec + pdf_diff), pdf_diff = 1.0f - pdf_spec;
        h = Unit(-r->wi + r->wo), h_local = r->Local(h);
        D = GgxPdfIso(alpha, h_local);
        const float h_o = Dot(r->wo, h);
        // float *= float / (double 4.0 * float): quotient and product in double
        pdf_spec = static_cast<float>(
            static_cast<double>(pdf_spec) *
            (static_cast<double>(D) / (4.0 * static_cast<double>(h_o))));
        pdf_diff *= pdf_local;
    }
    r->pdf = pdf_spec + pdf_diff;
    if (r->pdf < kEps)
        return;
    r->valid = true;
    if (pdf_spec > kEps)
    {
        const V3 li = r->Local(-r->wi), lo = r->Local(r->wo);
        const float h_i = Dot(-r->wi, h), F = Schlick(h_i, b.reflectivity),
                    G = (SmithG1Iso(alpha, lo, h_local) * SmithG1Iso(alpha, li, h_local));
        const V3 spec = V3((F * D * G) / (4.0f * n_o));
        r->attenuation += spec * ks;
    }
    if (pdf_diff > kEps)
    {
        V3 diff = kd * k1DivPi * n_i;
        diff *= ((1.0f - kr_i) * (1.0f - kr_o)) / (1.0f - b.f_avg);
        r->attenuation += diff;
    }
}

inline void PlasticEval(const Scene &sc, const Bsdf &b, Scatter *r) // plastic.cpp:98-153
{
    const float n_o = Dot(r->wo, r->normal);
    if (n_o < kEpsFloat)
        return;
    const V3 kd = TextureColor(sc, b.reflectance, r->uv),
             ks = TextureColor(sc, b.spec_r, r->uv);
    const float w_spec = (ks.x + ks.y + ks.z) / ((kd.x + kd.y + kd.z) + (ks.x + ks.y + ks.z));
    const float n_i = Dot(-r->wi, r->normal), kr_i = Schlick(n_i, b.reflectivity);
    float pdf_spec = kr_i * w_spec, pdf_diff = (1.0f - kr_i) * (1.0f - w_spec);
    pdf_spec = pdf_spec / (pdf_spec + pdf_diff);
    pdf_diff = 1.0f - pdf_spec;
    const V3 h = Unit(-r->wi + r->wo), h_local = r->Local(h);
    const float alpha = TextureColor(sc, b.roughness, r->uv).x,
                D = GgxPdfIso(alpha, h_local), h_o = Dot(r->wo, h);
    pdf_spec *= D / (4.0f * h_o);
    const V3 lo = r->Local(r->wo);
    pdf_diff *= lo.z;
    r->pdf = pdf_spec + pdf_diff;
    if (r->pdf < kEps)
        return;
    r->valid = true;
    if (pdf_spec > kEps)
    {
        const V3 li = r->Local(-r->wi);
        const float h_i = Dot(-r->wi, h), F = Schlick(h_i, b.reflectivity),
                    G = (SmithG1Iso(alpha, lo, h_local) * SmithG1Iso(alpha, li, h_local));
        const V3 spec = V3((F * D * G) / (4.0f * n_o));
        r->attenuation += spec * ks;
    }
    if (pdf_diff > kEps)
    {
        V3 diff = kd * k1DivPi * n_i;
        const float kr_o = Schlick(n_o, b.reflectivity);
        diff *= ((1.0f - kr_i) * (1.0f - kr_o)) / (1.0f - b.f_avg);
        r->attenuation += diff;
    }
}

// bsdf.cpp:188-236
inline void BsdfSample(const Scene &sc, const Bsdf &b, uint32_t *rng, Scatter *r)
{
    switch (b.type)
    {
    case Mat::kDiffuse: DiffuseSample(sc, b, rng, r); break;
    case Mat::kRoughDiffuse: RoughDiffuseSample(sc, b, rng, r); break;
    case Mat::kConductor: ConductorSample(sc, b, rng, r); break;
    case Mat::kDielectric: DielectricSample(sc, b, rng, r); break;
    case Mat::kThinDielectric: ThinDielectricSample(sc, b, rng, r); break;
    case Mat::kPlastic: PlasticSample(sc, b, rng, r); break;
    default: break;
    }
}

inline void BsdfEval(const Scene &sc, const Bsdf &b, Scatter *r)
{
    switch (b.type)
    {
    case Mat::kDiffuse: DiffuseEval(sc, b, r); break;
    case Mat::kRoughDiffuse: RoughDiffuseEval(sc, b, r); break;
    case Mat::kConductor: ConductorEval(sc, b, r); break;
    case Mat::kDielectric: DielectricEval(sc, b, r); break;
    case Mat::kThinDielectric: ThinDielectricEval(sc, b, r); break;
    case Mat::kPlastic: PlasticEval(sc, b, r); break;
    default: break;
    }
}

// ---------------------------------------------------------------------------
// Emitters (emitter.cpp:177-260 and the per-type files)
// ---------------------------------------------------------------------------
struct LightSample // emitter.hpp:49-55
{
    bool valid = false, harsh = true;
    float distance = kMaxF;
    V3 wi;
};

inline LightSample EmitterSample(const Scene &sc, const Emitter &e, V3 origin,
                                 float xi0, float xi1)
{
    LightSample s;
    switch (e.type)
    {
    case Light::kPoint: // point_light.cpp:8-19
    {
        const V3 d = origin - e.position;
        s = {true, true, Len(d), Unit(d)};
        break;
    }
    case Light::kSpot: // spot_light.cpp:8-24
    {
        const V3 d = origin - e.position;
        const V3 wi = Unit(d), local = XformDir(e.to_local, wi);
        if (local.z >= e.cos_cutoff)
            s = {true, true, Len(d), wi};
        break;
    }
    case Light::kDirectional: // directional_light.cpp:8-19
        s = {true, true, kMaxF, e.direction};
        break;
    case Light::kSun: // sun.cpp:8-19
        s = {true, true, kMaxF, FrameToWorld(ConeUniform(e.cos_cutoff, xi0, xi1), e.direction)};
        break;
    case Light::kEnvMap: // envmap.cpp:70-88 (tables mis-pointed on purpose, Q7)
    {
        const float *tab = sc.env_tables.data();
        const uint32_t row = CdfSearch(e.height + 1, tab + e.cdf_rows, xi0) - 1;
        const float *cdf_col = tab + e.cdf_cols + static_cast<size_t>(row) * (e.width + 1);
        const uint32_t col = CdfSearch(e.width + 1, cdf_col, xi1) - 1;
        const V3 local = FromSpherical(row * kPi / e.height, col * k2Pi / e.width, 1);
        s = {true, false, kMaxF, XformDir(e.to_world, local)};
        break;
    }
    case Light::kConstant: // constant_light.cpp:8-18
        s = {true, false, kMaxF, SphereUniform(xi0, xi1)};
        break;
    }
    return s;
}

inline V3 EnvLookup(const Scene &sc, const Emitter &e, V3 dir_local, V2 *uv_out)
{
    float phi = 0, theta = 0;
    ToSpherical(dir_local, &theta, &phi, nullptr);
    const V2 uv = {phi * k1Div2Pi, theta * k1DivPi};
    if (uv_out)
        *uv_out = uv;
    return TextureColor(sc, e.texture, uv);
}

// Radiance arriving along a sampled connection (emitter.cpp:206-231).
inline V3 EmitterEvalSample(const Scene &sc, const Emitter &e, const LightSample &s)
{
    switch (e.type)
    {
    case Light::kPoint: // point_light.cpp:21-25 — a point light contributes nothing
        return {};
    case Light::kSpot: // spot_light.cpp:26-44
    {
        const V3 d = XformDir(e.to_local, s.wi);
        V3 fall = {1.0f, 1.0f, 1.0f};
        if (e.texture != kNone)
        {
            const V2 uv = {0.5f + 0.5f * d.x / (d.z * e.uv_factor),
                           0.5f + 0.5f * d.y / (d.z * e.uv_factor)};
            fall *= TextureColor(sc, e.texture, uv);
        }
        if (d.z < e.cos_beam)
            fall *= (e.cutoff - acosf(d.z)) * e.transition_rcp;
        return e.intensity * fall * Sq(1.0f / s.distance);
    }
    case Light::kDirectional:
    case Light::kSun:
    case Light::kConstant:
        return e.radiance;
    case Light::kEnvMap: // envmap.cpp:90-98: looks up along -dir
        return EnvLookup(sc, e, -XformDir(e.to_local, s.wi), nullptr);
    }
    return {};
}

// Radiance seen along an escaping ray (emitter.cpp:233-248).
inline V3 EmitterEvalDir(const Scene &sc, const Emitter &e, V3 look)
{
    switch (e.type)
    {
    case Light::kSun: // sun.cpp:26-32
        return EnvLookup(sc, e, look, nullptr);
    case Light::kEnvMap: // envmap.cpp:100-107
        return EnvLookup(sc, e, XformDir(e.to_local, look), nullptr);
    case Light::kConstant:
        return e.radiance;
    default:
        return {};
    }
}

inline float EmitterPdf(const Scene &sc, const Emitter &e, V3 look) // emitter.cpp:250-261
{
    switch (e.type)
    {
    case Light::kEnvMap: // envmap.cpp:109-133 (row from texcoord.u, Q7)
    {
        const V3 d = XformDir(e.to_local, look);
        float phi = 0, theta = 0;
        ToSpherical(d, &theta, &phi, nullptr);
        const V2 uv = {phi * k1Div2Pi, theta * k1DivPi};
        const V3 c = TextureColor(sc, e.texture, uv);
        const float *wr = sc.env_tables.data() + e.weight_rows;
        const float row = fminf(fmaxf(uv.u * e.height, 0), e.height - 1);
        const int ri = static_cast<int>(row);
        const float t = row - ri;
        const float denom = fmaxf(static_cast<float>(fabs(static_cast<double>(sinf(theta)))), 1e-4f);
        if (t == 0)
            return Luminance(c) * wr[ri] * e.normalization / denom;
        return Luminance(c) * Mix(wr[ri], wr[ri + 1], t) * e.normalization / denom;
    }
    case Light::kConstant:
        return k1Div4Pi;
    default:
        return 0;
    }
}

// ---------------------------------------------------------------------------
// Participating media (medium.cpp, homogeneous.cpp, henyey_greenstein.cpp,
// isotropic.cpp).  log / exp are the double functions.
// ---------------------------------------------------------------------------
struct MediumSample // medium.hpp:55-62
{
    bool valid = false, scattered = false;
    float pdf = 1.0f, distance = 0;
    V3 attenuation = V3(1.0f);
};

inline float ExpNeg(float sigma, float d) // exp(-sigma * d) in double, to float on use
{
    return static_cast<float>(exp(static_cast<double>(-sigma * d)));
}

inline void MediumDistance(const Medium &m, float max_distance, uint32_t *rng,
                           MediumSample *r) // homogeneous.cpp:9-53
{
    float xi0 = Rand(rng);
    if (xi0 < m.sampling_weight)
    {
        xi0 /= m.sampling_weight;
        const int channel = static_cast<int>(Rand(rng) * 3);
        r->distance = static_cast<float>(
            -log(static_cast<double>(1.0f - xi0)) /
            static_cast<double>(m.sigma_t[channel]));
        if (r->distance < max_distance)
        {
            // pdf starts at its initial value 1 and is accumulated in place;
            // each term is float * double exp -> rounded into the float sum
            for (int d = 0; d < 3; ++d)
                r->pdf = static_cast<float>(
                    static_cast<double>(r->pdf) +
                    static_cast<double>(m.sigma_t[d]) *
                        exp(static_cast<double>(-m.sigma_t[d] * r->distance)));
            r->pdf *= m.sampling_weight * (1.0f / 3.0f);
            r->scattered = true;
        }
    }
    if (!r->scattered)
    {
        r->distance = max_distance;
        r->pdf = 0;
        for (int d = 0; d < 3; ++d)
            r->pdf = static_cast<float>(
                static_cast<double>(r->pdf) +
                exp(static_cast<double>(-m.sigma_t[d] * r->distance)));
        r->pdf = m.sampling_weight * (1.0f / 3.0f) * r->pdf + (1.0f - m.sampling_weight);
    }
    for (int d = 0; d < 3; ++d)
    {
        r->attenuation[d] = ExpNeg(m.sigma_t[d], r->distance);
        if (r->attenuation[d] > kEpsFloat)
            r->valid = true;
    }
    if (r->scattered)
        r->attenuation *= m.sigma_s;
}

inline void MediumTransmittance(const Medium &m, MediumSample *r) // homogeneous.cpp:55-81
{
    for (int d = 0; d < 3; ++d)
    {
        r->attenuation[d] = ExpNeg(m.sigma_t[d], r->distance);
        if (r->attenuation[d] > kEpsFloat)
            r->valid = true;
    }
    if (!r->valid)
        return;
    if (r->scattered)
    {
        for (int d = 0; d < 3; ++d)
            r->pdf += m.sigma_t[d] * r->attenuation[d];
        r->pdf *= m.sampling_weight * (1.0f / 3.0f);
        r->attenuation *= m.sigma_s;
    }
    else
    {
        for (int d = 0; d < 3; ++d)
            r->pdf += r->attenuation[d];
        r->pdf = m.sampling_weight * (1.0f / 3.0f) * r->pdf + (1.0f - m.sampling_weight);
    }
}

struct PhaseSample // medium.hpp:27-34
{
    bool valid = false;
    float pdf = 0;
    V3 wi, wo, attenuation;
};

inline void HgValue(V3 g, float cos_t, PhaseSample *r)
{
    const V3 temp = 1.0f + Sq(g) + 2.0f * cos_t * g;
    r->attenuation = k1Div4Pi * (1.0f - Sq(g)) / (temp * Sqrt3(temp));
    r->pdf = 0;
    for (int d = 0; d < 3; ++d)
        r->pdf += r->attenuation[d];
    r->pdf *= (1.0f / 3.0f);
}

inline void PhaseSampleDir(const Medium &m, uint32_t *rng, PhaseSample *r)
{
    if (!m.hg) // isotropic.cpp:9-15
    {
        r->valid = true;
        r->attenuation = V3(k1Div4Pi);
        r->pdf = k1Div4Pi;
        const float xi0 = Rand(rng), xi1 = Rand(rng);
        r->wi = SphereUniform(xi0, xi1);
        return;
    }
    // henyey_greenstein.cpp:9-43
    const int ch = static_cast<int>(Rand(rng) * 3);
    const float g = m.g[ch];
    float cos_t = 0;
    if (fabsf(g) < kEpsFloat)
    {
        cos_t = 1.0f - 2.0f * Rand(rng);
    }
    else
    {
        const float term = (1.0f - Sq(g)) / (1.0f - g + 2.0f * g * Rand(rng));
        cos_t = (1.0f + Sq(g) - Sq(term)) / (2.0f * g);
    }
    HgValue(m.g, cos_t, r);
    if (r->pdf < kEps)
        return;
    r->valid = true;
    const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - Sq(cos_t)));
    const float phi = k2Pi * Rand(rng);
    r->wi = {sin_t * cosf(phi), sin_t * sinf(phi), cos_t};
    r->wi = -FrameToWorld(r->wi, r->wo);
}

inline void PhaseEval(const Medium &m, PhaseSample *r)
{
    if (!m.hg) // isotropic.cpp:17-22
    {
        r->valid = true;
        r->attenuation = V3(k1Div4Pi);
        r->pdf = k1Div4Pi;
        return;
    }
    HgValue(m.g, Dot(-r->wi, r->wo), r); // henyey_greenstein.cpp:45-60
    if (r->pdf < kEps)
        return;
    r->valid = true;
}

} // namespace orc

#endif // ORACLE_SHADING_HPP
