// BSDF set of the renderer: diffuse, Oren-Nayar rough diffuse, GGX conductor,
// (rough) dielectric, thin dielectric, (rough) plastic, with Kulla-Conty
// multiple-scattering compensation from a 128x128 LUT.
//
// Behavioural contract = the reference CPU integrator
// (src/renderer/bsdfs/*.cpp), including its quirks: evaluation pdfs that are
// not the sampling pdfs (diffuse.cpp:12), roughness rescaling only when
// sampling dielectrics (dielectric.cpp:47-49), unsigned LUT range checks
// (kulla_conty.cpp:82-143), random numbers drawn left to right.
// Conventions: wo points away from the surface towards the previous vertex, wi
// points INTO the surface; `attenuation` includes cos(theta_i).
#ifndef MCPT_BSDFS_H
#define MCPT_BSDFS_H

#include "textures.h"

namespace mcpt
{

struct BsdfQuery
{
    bool valid, inside;
    float pdf;
    V2 uv;
    V3 wi, wo, normal, tangent, bitangent, attenuation;
};

MCPT_HD V3 to_local(const BsdfQuery &q, V3 v) // bsdf.cpp:100-103
{
    return normalize(V3{dot(v, q.tangent), dot(v, q.bitangent), dot(v, q.normal)});
}
MCPT_HD V3 to_world(const BsdfQuery &q, V3 v) // bsdf.cpp:105-108
{
    return normalize(v.x * q.tangent + v.y * q.bitangent + v.z * q.normal);
}

struct ShadeTables
{
    const TextureRec *textures;
    const float *texels;
    const float *lut_brdf, *lut_albedo;
    bool all_constant; // every texture is a constant (compile-time constant in the lean kernels)
};

MCPT_HD V3 tex(const ShadeTables &T, uint32_t id, V2 uv)
{
    return texture_color(T.textures, T.texels, id, uv, T.all_constant);
}

// ---- GGX --------------------------------------------------------------------
MCPT_HD void ggx_sample_aniso(float xi0, float xi1, float au, float av, V3 &h, float &pdf) // microfacet.cpp:21-35
{
    const float phi = (gl::atanf(av / au * gl::tanf(kPi + k2Pi * xi1)) + kPi * floorf(2.0f * xi1 + 0.5f));
    const float cos_p = gl::cosf(phi), sin_p = gl::sinf(phi), a2 = 1.0f / (sqr(cos_p / au) + sqr(sin_p / av));
    const float tan2 = static_cast<float>(D(a2 * xi0) / (1.0 - D(xi0)));
    const float cos_t = 1.0f / sqrtf(1.0f + tan2), sin_t = sqrtf(1.0f - sqr(cos_t));
    h = V3{sin_t * cos_p, sin_t * sin_p, cos_t};
    pdf = static_cast<float>(1.0 / (D(kPi * au * av) * pow3d(cos_t) * D(sqr(1.0f + tan2 / a2))));
}

MCPT_HD float ggx_pdf_iso(float alpha, V3 h) // microfacet.cpp:37-46
{
    const float c = h.z;
    if (c <= 0.0f)
        return 0.0f;
    const float c2 = sqr(c), tan2 = (1.0f - c2) / c2, c3 = static_cast<float>(pow3d(c)), a2 = sqr(alpha);
    return a2 / (kPi * c3 * sqr(a2 + tan2));
}

MCPT_HD float ggx_pdf_aniso(float au, float av, V3 h) // microfacet.cpp:48-59
{
    const float c = h.z;
    if (c <= 0.0f)
        return 0.0f;
    return c / (kPi * au * av * sqr(sqr(h.x / au) + sqr(h.y / av) + sqr(c)));
}

MCPT_HD float smith_g1_aniso(float au, float av, V3 v, V3 h) // microfacet.cpp:76-86
{
    if (v.z * h.z <= 0)
        return 0;
    const float tan2 = (sqr(au * v.x) + sqr(av * v.y)) / sqr(v.z);
    return 2.0f / (1.0f + sqrtf(1.0f + tan2));
}

MCPT_HD float schlick_weight(float cos_t) { return static_cast<float>(pow5d(1.0f - cos_t)); } // microfacet.hpp:24-29
MCPT_HD float schlick(float cos_t, float r) { return (1.0f - r) * schlick_weight(cos_t) + r; }
MCPT_HD V3 schlick(float cos_t, V3 r) { return (1.0f - r) * schlick_weight(cos_t) + r; }

// ---- Kulla-Conty lookups (kulla_conty.cpp:82-143) ---------------------------
// The reference compares the truncated int indices with an UNSIGNED constant,
// so negative indices (negative cosines reach here from transmitted
// connections, dielectric.cpp:211-212) take the clamped branch.
MCPT_HD float lut_directional(const ShadeTables &T, float cos_t, float alpha)
{
    const float *t = T.lut_brdf;
    const float o1 = alpha * kLutRes, o2 = cos_t * kLutRes;
    const int i1 = static_cast<int>(o1), i2 = static_cast<int>(o2);
    constexpr uint32_t last = kLutRes - 1;
    const bool row_clamped = static_cast<uint32_t>(i1) >= last, col_clamped = static_cast<uint32_t>(i2) >= last;
    if (row_clamped)
    {
        if (col_clamped)
            return t[last * kLutRes + last];
        return lerp(t[last * kLutRes + i2], t[last * kLutRes + i2 + 1], o2 - i2);
    }
    if (col_clamped)
        return lerp(t[i1 * kLutRes + last], t[(i1 + 1) * kLutRes + last], o1 - i1);
    return lerp(lerp(t[i1 * kLutRes + i2], t[(i1 + 1) * kLutRes + i2], o1 - i1),
                lerp(t[i1 * kLutRes + i2 + 1], t[(i1 + 1) * kLutRes + i2 + 1], o1 - i1), o2 - i2);
}

MCPT_HD float lut_average(const ShadeTables &T, float alpha)
{
    const float o = alpha * kLutRes;
    const int i = static_cast<int>(o);
    if (static_cast<uint32_t>(i) >= static_cast<uint32_t>(kLutRes - 1))
        return T.lut_albedo[kLutRes - 1];
    return lerp(T.lut_albedo[i], T.lut_albedo[i + 1], o - i);
}

// ---- diffuse (diffuse.cpp) --------------------------------------------------
MCPT_HD void diffuse_eval(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q)
{
    q.pdf = dot(q.wo, q.normal); // outgoing cosine, no 1/pi: reference quirk
    if (q.pdf < kEps)
        return;
    q.valid = true;
    q.attenuation = tex(T, b.tex0, q.uv) * k1DivPi * dot(-q.wi, q.normal);
}

MCPT_HD void diffuse_sample(const ShadeTables &T, const BsdfRec &b, uint32_t &rng, BsdfQuery &q)
{
    const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
    V3 local;
    sample_hemisphere_cosine(xi0, xi1, local, q.pdf);
    if (q.pdf < kEps)
        return;
    q.wi = -to_world(q, local);
    q.valid = true;
    q.attenuation = tex(T, b.tex0, q.uv) * k1DivPi * local.z;
}

// ---- Oren-Nayar (rough_diffuse.cpp:10-95, always the full model) ------------
MCPT_HD void oren_nayar(float rough, V3 albedo, BsdfQuery &q)
{
    constexpr float conv = 0.70710678118f;
    const float s2 = sqr(rough * conv);
    const V3 li = to_local(q, -q.wi), lo = to_local(q, q.wo);
    const float n_i = li.z, n_o = lo.z, sin_i = sqrtf(1.0f - n_i * n_i), sin_o = sqrtf(1.0f - n_o * n_o);
    float phi_i, theta_i, phi_o, theta_o;
    to_spherical(li, theta_i, phi_i);
    to_spherical(lo, theta_o, phi_o);
    const float cos_dphi = gl::cosf(phi_i) * gl::cosf(phi_o) + gl::sinf(phi_i) * gl::sinf(phi_o);
    const float alpha = fmaxf(theta_i, theta_o), beta = fminf(theta_i, theta_o);
    float sin_a, sin_b, tan_b;
    if (n_i > n_o)
        sin_a = sin_o, sin_b = sin_i, tan_b = sin_i / n_i;
    else
        sin_a = sin_i, sin_b = sin_o, tan_b = sin_o / n_o;
    const float tmp = s2 / (s2 + 0.09f), tmp2 = 4.0f * k1DivPi * k1DivPi * alpha * beta,
                tmp3 = 2.0f * beta * k1DivPi;
    const float C1 = 1.0f - 0.5f * s2 / (s2 + 0.33f);
    float C2 = 0.45f * tmp;
    const float C3 = 0.125f * tmp * tmp2 * tmp2, C4 = 0.17f * s2 / (s2 + 0.13f);
    if (cos_dphi > 0)
        C2 *= sin_a;
    else
        C2 = static_cast<float>(D(C2) * (D(sin_a) - pow3d(tmp3)));
    const float tan_half = static_cast<float>(
        D(sin_a + sin_b) / (D(sqrtf(fmaxf(0.0f, 1.0f - sqr(sin_a)))) + sqrt(D(fmaxf(0.0f, 1.0f - sqr(sin_b))))));
    const double single = D(C1 + cos_dphi * C2 * tan_b) + (1.0 - fabs(D(cos_dphi))) * D(C3) * D(tan_half);
    const V3 sngl = albedo * static_cast<float>(single), dbl = sqr(albedo) * (C4 * (1.0f - cos_dphi * sqr(tmp3)));
    q.attenuation = (sngl + dbl) * k1DivPi * n_i;
}

MCPT_HD void rough_diffuse_eval(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q) // rough_diffuse.cpp:117-128
{
    q.pdf = dot(q.wo, q.normal);
    if (q.pdf < kEps)
        return;
    q.valid = true;
    const float alpha = tex(T, b.tex1, q.uv).x;
    oren_nayar(alpha, tex(T, b.tex0, q.uv), q);
}

MCPT_HD void rough_diffuse_sample(const ShadeTables &T, const BsdfRec &b, uint32_t &rng, BsdfQuery &q) // :99-115
{
    const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
    V3 local;
    sample_hemisphere_cosine(xi0, xi1, local, q.pdf);
    if (q.pdf < kEps)
        return;
    q.wi = -normalize(local.x * q.tangent + local.y * q.bitangent + local.z * q.normal);
    q.valid = true;
    const float alpha = tex(T, b.tex1, q.uv).x;
    oren_nayar(alpha, tex(T, b.tex0, q.uv), q);
}

// ---- conductor (conductor.cpp) ----------------------------------------------
MCPT_HD V3 conductor_multiscatter(const ShadeTables &T, const BsdfRec &b, float n_i, float n_o, float alpha)
{
    const float e_i = lut_directional(T, n_i, alpha), e_o = lut_directional(T, n_o, alpha),
                e_avg = lut_average(T, alpha), f_ms = (1.0f - e_i) * (1.0f - e_o) / (kPi * (1.0f - e_avg));
    const V3 favg = from(b.f_avg3);
    const V3 f_add = sqr(favg) * e_avg / (1.0f - favg * (1.0f - e_avg));
    return f_ms * f_add * n_i;
}

MCPT_HD void conductor_sample(const ShadeTables &T, const BsdfRec &b, uint32_t &rng, BsdfQuery &q) // :34-77
{
    const float au = tex(T, b.tex0, q.uv).x, av = tex(T, b.tex1, q.uv).x;
    const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
    V3 h_local;
    float Dh;
    ggx_sample_aniso(xi0, xi1, au, av, h_local, Dh);
    const V3 h = to_world(q, h_local);
    const float h_o = dot(q.wo, h);
    q.pdf = Dh / (4.0f * h_o);
    if (q.pdf < kEps)
        return;
    q.wi = -reflect(-q.wo, h);
    const float n_i = dot(-q.wi, q.normal);
    if (n_i < kEpsFloat)
        return;
    q.valid = true;
    const V3 li = to_local(q, -q.wi), lo = to_local(q, q.wo);
    const float G = smith_g1_aniso(au, av, li, h_local) * smith_g1_aniso(au, av, lo, h_local),
                h_i = dot(-q.wi, h), n_o = lo.z;
    q.attenuation = (schlick(h_i, from(b.reflectivity3)) * Dh * G) / (4.0f * n_o);
    if (au == av)
        q.attenuation += conductor_multiscatter(T, b, n_i, n_o, au);
    q.attenuation *= tex(T, b.tex2, q.uv);
}

MCPT_HD void conductor_eval(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q) // :79-119
{
    const float n_o = dot(q.wo, q.normal);
    if (n_o < kEpsFloat)
        return;
    const V3 h = normalize(-q.wi + q.wo), h_local = to_local(q, h);
    const float au = tex(T, b.tex0, q.uv).x, av = tex(T, b.tex1, q.uv).x,
                Dh = ggx_pdf_aniso(au, av, h_local), h_o = dot(q.wo, h);
    q.pdf = Dh / (4.0f * h_o);
    if (q.pdf < kEps)
        return;
    q.valid = true;
    const V3 li = to_local(q, -q.wi), lo = to_local(q, q.wo);
    const float G = smith_g1_aniso(au, av, li, h_local) * smith_g1_aniso(au, av, lo, h_local),
                h_i = dot(-q.wi, h);
    q.attenuation = (schlick(h_i, from(b.reflectivity3)) * Dh * G) / (4.0f * n_o);
    if (au == av)
        q.attenuation += conductor_multiscatter(T, b, dot(-q.wi, q.normal), n_o, au);
    q.attenuation *= tex(T, b.tex2, q.uv);
}

// ---- dielectric (dielectric.cpp) --------------------------------------------
// dielectric.cpp:30-33: (1 - F_avg)(1 - F_avg_inv) eta^2 / ((1 - F_avg) + (1 - F_avg_inv) eta^2), in double, rounded to float.
// host/commit.cpp stores it for eta (seen from outside) and 1 / eta (from inside) in the record.
MCPT_HD float dielectric_ms_ratio_t(const BsdfRec &b, float eta)
{
    const double eta2 = pow2d(eta);
    return static_cast<float>(D((1.0f - b.f_avg) * (1.0f - b.f_avg_inv)) * eta2 / (D(1.0f - b.f_avg) + D(1.0f - b.f_avg_inv) * eta2));
}

MCPT_HD float dielectric_multiscatter(const ShadeTables &T, const BsdfRec &b, float n_i, float n_o, float alpha,
                                      bool inside, bool reflected) // :14-38
{
    const float e_i = lut_directional(T, n_i, alpha), e_o = lut_directional(T, n_o, alpha),
                e_avg = lut_average(T, alpha), f_ms = (1.0f - e_i) * (1.0f - e_o) / (kPi * (1.0f - e_avg));
    const float F = inside ? b.f_avg_inv : b.f_avg;
    const float f_add = static_cast<float>(pow2d(F) * D(e_avg) / D(1.0f - F * (1.0f - e_avg)));
    const float ratio_t = inside ? b.ms_ratio_t_inside : b.ms_ratio_t; // (dielectric_ms_ratio_t(b, eta), evaluated at commit)
    const float ret = f_ms * f_add * n_i;
    return reflected ? (1.0f - ratio_t) * ret : ratio_t * ret;
}

MCPT_HD void dielectric_sample(const ShadeTables &T, const BsdfRec &b, uint32_t &rng, BsdfQuery &q) // :44-140
{
    const float scale = static_cast<float>(D(1.2f) - D(0.2f) * sqrt(D(fabsf(dot(-q.wo, q.normal)))));
    const float au = tex(T, b.tex0, q.uv).x * scale, av = tex(T, b.tex1, q.uv).x * scale;
    const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
    V3 h_local;
    float Dh;
    ggx_sample_aniso(xi0, xi1, au, av, h_local, Dh);
    const V3 h = to_world(q, h_local);
    const float h_o = dot(q.wo, h);
    if (h_o < kEpsFloat)
        return;
    float eta = b.eta, eta_inv = b.eta_inv;
    if (!q.inside)
    {
        const float t = eta_inv;
        eta_inv = eta;
        eta = t;
    }
    V3 wt = V3{0, 0, 0};
    const bool total_reflection = !refract(-q.wo, h, eta, wt);
    const float F = schlick(h_o, b.reflectivity);
    const V3 lo = to_local(q, q.wo);
    // dielectric.cpp:80-138 continues in two arms, reflection and refraction.  The lanes of a wavefront take both, so what the arms
    // share — the incident direction's local frame, both shadowing terms, the multiple-scattering term with its table lookups, the
    // texture — is written ONCE here, fed with the lane's arm's operands: the same operations per lane, in one pass of the wavefront.
    const bool reflected = total_reflection || lcg_next(rng) < F; // (no draw after a total reflection)
    q.wi = reflected ? -reflect(-q.wo, h) : -wt;
    V3 li = to_local(q, -q.wi);
    float n_i, h_i = 0.0f, F_t = 0.0f, h_o_t = 0.0f;
    if (reflected)
        n_i = dot(-q.wi, q.normal);
    else
    {
        li.z = -li.z;
        n_i = li.z;
        h_i = -dot(wt, h);
    }
    if (n_i < kEpsFloat || (!reflected && h_i < kEpsFloat))
        return;
    if (reflected)
        q.pdf = F * Dh / (4.0f * h_o);
    else
    {
        h_o_t = -h_o;
        F_t = schlick(h_i, b.reflectivity);
        q.pdf = ((1.0f - F_t) * Dh) * fabsf(h_o_t / sqr(eta_inv * h_i + h_o_t));
    }
    if (q.pdf < kEps)
        return;
    const float G = smith_g1_aniso(au, av, li, h_local) * smith_g1_aniso(au, av, lo, h_local), n_o = lo.z;
    float a = reflected ? (F * Dh * G) / (4.0f * n_o)
                        : ((fabsf(h_i) * fabsf(h_o_t)) * ((1.0f - F_t) * G * Dh)) / fabsf(n_o * sqr(eta_inv * h_i + h_o_t));
    if (au == av)
        a += dielectric_multiscatter(T, b, n_i, n_o, au, reflected ? q.inside : !q.inside, reflected);
    const V3 weight = reflected ? splat(a) : splat(a) * sqr(eta);
    q.attenuation = weight * tex(T, reflected ? b.tex2 : b.tex3, q.uv);
    q.valid = true;
}

MCPT_HD void dielectric_eval(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q) // :142-224
{
    float eta = b.eta, eta_inv = b.eta_inv;
    if (q.inside)
    {
        const float t = eta_inv;
        eta_inv = eta;
        eta = t;
    }
    const float n_o = dot(q.wo, q.normal);
    const bool reflected = n_o > 0.0f;
    const V3 h = reflected ? normalize(-q.wi + q.wo) : -normalize(eta_inv * (-q.wi) + q.wo), h_local = to_local(q, h);
    const float au = tex(T, b.tex0, q.uv).x, av = tex(T, b.tex1, q.uv).x, Dh = ggx_pdf_aniso(au, av, h_local),
                h_i = dot(-q.wi, h), h_o = dot(q.wo, h), F = schlick(h_i, b.reflectivity);
    q.pdf = reflected ? (F * Dh) / (4.0f * h_o) : (((1.0f - F) * Dh) * fabsf(h_o / sqr(eta_inv * h_i + h_o)));
    if (q.pdf < kEps)
        return;
    q.valid = true;
    // (dielectric.cpp:186-222 in two arms; what they share is evaluated once per wavefront, as in dielectric_sample)
    const V3 li = to_local(q, -q.wi);
    const V3 lo = to_local(q, reflected ? q.wo : -q.wo);
    const float G = smith_g1_aniso(au, av, li, h_local) * smith_g1_aniso(au, av, lo, h_local);
    float a = reflected ? (F * Dh * G) / (4.0f * n_o) : ((fabsf(h_i) * fabsf(h_o)) * ((1.0f - F) * G * Dh)) / fabsf(n_o * sqr(eta_inv * h_i + h_o));
    if (au == av)
        a += dielectric_multiscatter(T, b, dot(-q.wi, q.normal), n_o, au, q.inside, reflected);
    const V3 weight = reflected ? splat(a) : splat(a) * sqr(eta);
    q.attenuation = weight * tex(T, reflected ? b.tex2 : b.tex3, q.uv);
}

// ---- thin dielectric (thin_dielectric.cpp) ----------------------------------
MCPT_HD void thin_dielectric_sample(const ShadeTables &T, const BsdfRec &b, uint32_t &rng, BsdfQuery &q) // :11-67
{
    const float au = tex(T, b.tex0, q.uv).x, av = tex(T, b.tex1, q.uv).x;
    const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
    V3 h_local;
    float Dh;
    ggx_sample_aniso(xi0, xi1, au, av, h_local, Dh);
    const V3 h = to_world(q, h_local);
    const float h_o = dot(q.wo, h);
    q.pdf = Dh / (4.0f * h_o);
    if (q.pdf < kEps)
        return;
    q.wi = -reflect(-q.wo, h);
    if (dot(-q.wi, q.normal) < kEpsFloat)
        return;
    const V3 li = to_local(q, -q.wi), lo = to_local(q, q.wo);
    const float G = smith_g1_aniso(au, av, li, h_local) * smith_g1_aniso(au, av, lo, h_local),
                h_i = dot(-q.wi, h), n_o = lo.z;
    float F = schlick(h_i, b.reflectivity);
    if (F < 1.0f)
        F *= 2.0f / (1.0f + F);
    if (lcg_next(rng) < F)
    {
        q.pdf *= F;
        if (q.pdf < kEps)
            return;
        q.attenuation = splat((F * Dh * G) / (4.0f * n_o)) * tex(T, b.tex2, q.uv);
    }
    else
    {
        q.pdf *= 1.0f - F;
        if (q.pdf < kEps)
            return;
        q.attenuation = splat(((1.0f - F) * Dh * G) / (4.0f * n_o)) * tex(T, b.tex3, q.uv);
        q.wi = q.wo;
    }
    q.valid = true;
}

MCPT_HD void thin_dielectric_eval(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q) // :69-124
{
    bool reflected = true;
    V3 wo = q.wo;
    float n_o = dot(q.wo, q.normal);
    if (fabsf(n_o) < kEpsFloat)
        return;
    V3 lo = to_local(q, q.wo);
    if (n_o < 0.0f)
    {
        reflected = false;
        n_o = -n_o;
        lo.z = -lo.z;
        wo = to_world(q, lo);
    }
    const V3 h = normalize(-q.wi + wo), h_local = to_local(q, h);
    const float au = tex(T, b.tex0, q.uv).x, av = tex(T, b.tex1, q.uv).x, Dh = ggx_pdf_aniso(au, av, h_local),
                h_i = dot(-q.wi, h), h_o = dot(q.wo, h);
    float F = schlick(h_i, b.reflectivity);
    if (F < 1.0f)
        F *= 2.0f / (1.0f + F);
    q.pdf = reflected ? (F * Dh) / (4.0f * h_o) : ((1.0f - F) * Dh) / (4.0f * h_o);
    if (q.pdf < kEps)
        return;
    q.valid = true;
    const V3 li = to_local(q, -q.wi);
    const float G = smith_g1_aniso(au, av, li, h_local) * smith_g1_aniso(au, av, lo, h_local);
    if (reflected)
        q.attenuation = splat((F * Dh * G) / (4.0f * n_o)) * tex(T, b.tex2, q.uv);
    else
        q.attenuation = splat(((1.0f - F) * Dh * G) / (4.0f * n_o)) * tex(T, b.tex3, q.uv);
}

// ---- plastic (plastic.cpp) --------------------------------------------------
MCPT_HD void plastic_lobes(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q, V3 kd, V3 ks, V3 h, V3 h_local,
                           float alpha, float Dh, float n_i, float n_o, float kr_i, float kr_o, float pdf_spec,
                           float pdf_diff)
{
    q.attenuation = V3{0, 0, 0};
    if (pdf_spec > kEps)
    {
        const V3 li = to_local(q, -q.wi), lo = to_local(q, q.wo);
        const float h_i = dot(-q.wi, h), F = schlick(h_i, b.reflectivity),
                    G = (smith_g1_iso(alpha, lo, h_local) * smith_g1_iso(alpha, li, h_local));
        q.attenuation += splat((F * Dh * G) / (4.0f * n_o)) * ks;
    }
    if (pdf_diff > kEps)
    {
        V3 diff = kd * k1DivPi * n_i;
        diff *= ((1.0f - kr_i) * (1.0f - kr_o)) / (1.0f - b.f_avg);
        q.attenuation += diff;
    }
}

MCPT_HD void plastic_sample(const ShadeTables &T, const BsdfRec &b, uint32_t &rng, BsdfQuery &q) // :11-96
{
    const V3 kd = tex(T, b.tex1, q.uv), ks = tex(T, b.tex2, q.uv);
    const float w_spec = (ks.x + ks.y + ks.z) / ((kd.x + kd.y + kd.z) + (ks.x + ks.y + ks.z));
    const float n_o = dot(q.wo, q.normal), kr_o = schlick(n_o, b.reflectivity);
    float kr_i = kr_o, pdf_spec = kr_i * w_spec, pdf_diff = (1.0f - kr_i) * (1.0f - w_spec);
    pdf_spec = pdf_spec / (pdf_spec + pdf_diff);
    pdf_diff = 1.0f - pdf_spec;
    V3 h_local = V3{0, 0, 0}, h = V3{0, 0, 0};
    float Dh = 0, n_i = 0;
    const float alpha = tex(T, b.tex0, q.uv).x;
    if (lcg_next(rng) < pdf_spec)
    {
        const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
        ggx_sample_iso(xi0, xi1, alpha, h_local, Dh);
        h = to_world(q, h_local);
        q.wi = -reflect(-q.wo, h);
        n_i = dot(-q.wi, q.normal);
        if (n_i < kEpsFloat)
            return;
        kr_i = schlick(n_i, b.reflectivity);
        pdf_spec = kr_i * w_spec, pdf_diff = (1.0f - kr_i) * w_spec;
        pdf_spec = pdf_spec / (pdf_spec + pdf_diff), pdf_diff = 1.0f - pdf_spec;
        pdf_spec *= Dh / (4.0f * dot(q.wo, h));
        pdf_diff *= dot(-q.wi, q.normal);
    }
    else
    {
        const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
        V3 local;
        float pdf_local;
        sample_hemisphere_cosine(xi0, xi1, local, pdf_local);
        q.wi = -to_world(q, local);
        n_i = dot(-q.wi, q.normal);
        kr_i = schlick(n_i, b.reflectivity);
        pdf_spec = kr_i * w_spec, pdf_diff = (1.0f - kr_i) * w_spec;
        pdf_spec = pdf_spec / (pdf_spec + pdf_diff), pdf_diff = 1.0f - pdf_spec;
        h = normalize(-q.wi + q.wo), h_local = to_local(q, h);
        Dh = ggx_pdf_iso(alpha, h_local);
        pdf_spec = static_cast<float>(D(pdf_spec) * (D(Dh) / (4.0 * D(dot(q.wo, h)))));
        pdf_diff *= pdf_local;
    }
    q.pdf = pdf_spec + pdf_diff;
    if (q.pdf < kEps)
        return;
    q.valid = true;
    plastic_lobes(T, b, q, kd, ks, h, h_local, alpha, Dh, n_i, n_o, kr_i, kr_o, pdf_spec, pdf_diff);
}

MCPT_HD void plastic_eval(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q) // :98-153
{
    const float n_o = dot(q.wo, q.normal);
    if (n_o < kEpsFloat)
        return;
    const V3 kd = tex(T, b.tex1, q.uv), ks = tex(T, b.tex2, q.uv);
    const float w_spec = (ks.x + ks.y + ks.z) / ((kd.x + kd.y + kd.z) + (ks.x + ks.y + ks.z));
    const float n_i = dot(-q.wi, q.normal), kr_i = schlick(n_i, b.reflectivity);
    float pdf_spec = kr_i * w_spec, pdf_diff = (1.0f - kr_i) * (1.0f - w_spec);
    pdf_spec = pdf_spec / (pdf_spec + pdf_diff);
    pdf_diff = 1.0f - pdf_spec;
    const V3 h = normalize(-q.wi + q.wo), h_local = to_local(q, h);
    const float alpha = tex(T, b.tex0, q.uv).x, Dh = ggx_pdf_iso(alpha, h_local);
    pdf_spec *= Dh / (4.0f * dot(q.wo, h));
    pdf_diff *= to_local(q, q.wo).z;
    q.pdf = pdf_spec + pdf_diff;
    if (q.pdf < kEps)
        return;
    q.valid = true;
    const float kr_o = schlick(n_o, b.reflectivity);
    plastic_lobes(T, b, q, kd, ks, h, h_local, alpha, Dh, n_i, n_o, kr_i, kr_o, pdf_spec, pdf_diff);
}

// ---- dispatch (bsdf.cpp:188-236) --------------------------------------------
// kOnly: 0 = whatever kind the record says (the single-kernel formulations); a BsdfKind = the caller guarantees that
// kind (the queued renderer's per-material shade launches, hip/queued_kernels.hip: only that model is compiled in);
// kBsdfNoCode = the caller never reaches a BSDF (its launch shades misses, emitters and pass-through surfaces).
constexpr uint32_t kBsdfNoCode = 0xFFu;
// Developer switch for register-pressure experiments: -DMCPT_BSDF_KINDS=<bit per BsdfKind> compiles the dispatch for
// those kinds only (the default compiles all of them).
#ifndef MCPT_BSDF_KINDS
#define MCPT_BSDF_KINDS 0xFFFFFFFFu
#endif
MCPT_HD constexpr bool bsdf_kind_compiled(uint32_t kind) { return ((MCPT_BSDF_KINDS >> kind) & 1u) != 0; }

// kKinds: bit per BsdfKind the instantiation compiles — the caller guarantees that the scene has no other (device_scene.h:
// kFeatNoTransmission, kFeatDielectricOnly, kFeatConductorOnly; path_core.h, Config::kKinds).
template <bool kMicrofacet, uint32_t kOnly = 0, uint32_t kKinds = 0xFFFFFFFFu>
MCPT_HD void bsdf_sample(const ShadeTables &T, const BsdfRec &b, uint32_t &rng, BsdfQuery &q)
{
    if (kOnly == kBsdfNoCode)
        return;
    const uint32_t kind = kOnly ? kOnly : b.kind;
    if (!kMicrofacet || kind == kBsdfDiffuse)
    {
        if (kind == kBsdfDiffuse)
            diffuse_sample(T, b, rng, q);
        return;
    }
    switch (kind)
    {
    case kBsdfRoughDiffuse: if (((kKinds >> kBsdfRoughDiffuse) & 1u) && bsdf_kind_compiled(kBsdfRoughDiffuse)) rough_diffuse_sample(T, b, rng, q); break;
    case kBsdfConductor: if (((kKinds >> kBsdfConductor) & 1u) && bsdf_kind_compiled(kBsdfConductor)) conductor_sample(T, b, rng, q); break;
    case kBsdfDielectric: if (((kKinds >> kBsdfDielectric) & 1u) && bsdf_kind_compiled(kBsdfDielectric)) dielectric_sample(T, b, rng, q); break;
    case kBsdfThinDielectric: if (((kKinds >> kBsdfThinDielectric) & 1u) && bsdf_kind_compiled(kBsdfThinDielectric)) thin_dielectric_sample(T, b, rng, q); break;
    case kBsdfPlastic: if (((kKinds >> kBsdfPlastic) & 1u) && bsdf_kind_compiled(kBsdfPlastic)) plastic_sample(T, b, rng, q); break;
    default: break;
    }
}

template <bool kMicrofacet, uint32_t kOnly = 0, uint32_t kKinds = 0xFFFFFFFFu>
MCPT_HD void bsdf_eval(const ShadeTables &T, const BsdfRec &b, BsdfQuery &q)
{
    if (kOnly == kBsdfNoCode)
        return;
    const uint32_t kind = kOnly ? kOnly : b.kind;
    if (!kMicrofacet || kind == kBsdfDiffuse)
    {
        if (kind == kBsdfDiffuse)
            diffuse_eval(T, b, q);
        return;
    }
    switch (kind)
    {
    case kBsdfRoughDiffuse: if (((kKinds >> kBsdfRoughDiffuse) & 1u) && bsdf_kind_compiled(kBsdfRoughDiffuse)) rough_diffuse_eval(T, b, q); break;
    case kBsdfConductor: if (((kKinds >> kBsdfConductor) & 1u) && bsdf_kind_compiled(kBsdfConductor)) conductor_eval(T, b, q); break;
    case kBsdfDielectric: if (((kKinds >> kBsdfDielectric) & 1u) && bsdf_kind_compiled(kBsdfDielectric)) dielectric_eval(T, b, q); break;
    case kBsdfThinDielectric: if (((kKinds >> kBsdfThinDielectric) & 1u) && bsdf_kind_compiled(kBsdfThinDielectric)) thin_dielectric_eval(T, b, q); break;
    case kBsdfPlastic: if (((kKinds >> kBsdfPlastic) & 1u) && bsdf_kind_compiled(kBsdfPlastic)) plastic_eval(T, b, q); break;
    default: break;
    }
}

} // namespace mcpt

#endif // MCPT_BSDFS_H
