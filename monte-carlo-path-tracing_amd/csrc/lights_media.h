// Emitters (point, spot, directional, sun cone, lat-long environment map,
// constant environment) and participating media (homogeneous medium with
// isotropic or Henyey-Greenstein phase function).
//
// Behavioural contract = the reference CPU integrator:
// src/renderer/emitters/*.cpp, src/renderer/medium/*.cpp — including the
// env-map table pointer mix-up (emitter.cpp:166-175), the
// "harsh" flag that disables MIS for delta lights, the free-flight pdf that is
// accumulated on top of its initial value 1 (homogeneous.cpp:20-27), and the
// double-precision log/exp calls.
#ifndef MCPT_LIGHTS_MEDIA_H
#define MCPT_LIGHTS_MEDIA_H

#include "textures.h"

namespace mcpt
{

struct LightSample // emitter.hpp:49-55
{
    bool harsh;
    float distance;
    V3 wi;
};

struct LightTables
{
    const TextureRec *textures;
    const float *texels;
    const float *env_tables;
    bool all_constant; // see ShadeTables
};

MCPT_HD V3 latlong_lookup(const LightTables &T, uint32_t texture, V3 dir, float &theta, V2 &uv)
{
    float phi;
    to_spherical(dir, theta, phi);
    uv = V2{phi * k1Div2Pi, theta * k1DivPi};
    return texture_color(T.textures, T.texels, texture, uv, T.all_constant);
}

// emitter.cpp:177-204.  xi0/xi1 are always drawn by the caller, also for the
// lights that ignore them.
MCPT_HD LightSample emitter_sample(const LightTables &T, const EmitterRec &e, V3 origin, float xi0, float xi1)
{
    LightSample s;
    s.harsh = true, s.distance = kMaxFloat, s.wi = V3{0, 0, 0};
    switch (e.kind)
    {
    case kEmitPoint: // point_light.cpp:8-19
    {
        const V3 d = origin - from(e.position);
        s.distance = length(d), s.wi = normalize(d);
        break;
    }
    case kEmitSpot: // spot_light.cpp:8-24: outside the cone the record stays at its defaults
    {
        const V3 d = origin - from(e.position), wi = normalize(d);
        if (transform_dir(e.to_local, wi).z >= e.cos_cutoff)
            s.distance = length(d), s.wi = wi;
        break;
    }
    case kEmitDirectional: // directional_light.cpp:8-19
        s.wi = from(e.direction);
        break;
    case kEmitSun: // sun.cpp:8-19
        s.wi = frame_to_world(sample_cone_uniform(e.cos_cutoff, xi0, xi1), from(e.direction));
        break;
    case kEmitEnvMap: // envmap.cpp:70-88
    {
        const uint32_t row = cdf_search_long(e.height + 1, T.env_tables + e.cdf_rows, xi0) - 1;
        const uint32_t col = cdf_search_long(e.width + 1, T.env_tables + e.cdf_cols + row * (e.width + 1), xi1) - 1;
        s.harsh = false;
        s.wi = transform_dir(e.to_world, from_spherical(row * kPi / e.height, col * k2Pi / e.width, 1.0f));
        break;
    }
    case kEmitConstant: // constant_light.cpp:8-18
        s.harsh = false;
        s.wi = sample_sphere_uniform(xi0, xi1);
        break;
    default:
        break;
    }
    return s;
}

// Radiance carried by a sampled connection (emitter.cpp:206-231).
MCPT_HD V3 emitter_eval_sample(const LightTables &T, const EmitterRec &e, const LightSample &s)
{
    switch (e.kind)
    {
    case kEmitSpot: // spot_light.cpp:26-44
    {
        const V3 d = transform_dir(e.to_local, s.wi);
        V3 fall = V3{1.0f, 1.0f, 1.0f};
        if (e.texture != kNone)
        {
            const V2 uv = V2{0.5f + 0.5f * d.x / (d.z * e.uv_factor), 0.5f + 0.5f * d.y / (d.z * e.uv_factor)};
            fall *= texture_color(T.textures, T.texels, e.texture, uv, T.all_constant);
        }
        if (d.z < e.cos_beam)
            fall *= (e.cutoff - gl::acosf(d.z)) * e.transition_rcp;
        return from(e.intensity) * fall * sqr(1.0f / s.distance);
    }
    case kEmitDirectional:
    case kEmitSun:
    case kEmitConstant:
        return from(e.radiance);
    case kEmitEnvMap: // envmap.cpp:90-98: looked up along -dir
    {
        float theta;
        V2 uv;
        return latlong_lookup(T, e.texture, -transform_dir(e.to_local, s.wi), theta, uv);
    }
    default: // a point light contributes nothing (point_light.cpp:21-25)
        return V3{0, 0, 0};
    }
}

// Radiance seen along an escaping ray (emitter.cpp:233-248).
MCPT_HD V3 emitter_eval_dir(const LightTables &T, const EmitterRec &e, V3 look)
{
    float theta;
    V2 uv;
    switch (e.kind)
    {
    case kEmitSun: // sun.cpp:26-32
        return latlong_lookup(T, e.texture, look, theta, uv);
    case kEmitEnvMap: // envmap.cpp:100-107
        return latlong_lookup(T, e.texture, transform_dir(e.to_local, look), theta, uv);
    case kEmitConstant:
        return from(e.radiance);
    default:
        return V3{0, 0, 0};
    }
}

// Environment map: the radiance along a direction (given in the map's frame) and the pdf of having sampled it, from ONE lookup.
// envmap.cpp:90-133 looks the texel up once for the radiance and once more inside the pdf — the same angles, the same four
// texels; the callers that need both (an escaped ray's MIS weight, a sampled connection's) ask here.
MCPT_HD V3 envmap_eval_pdf(const LightTables &T, const EmitterRec &e, V3 local_dir, float &pdf)
{
    // envmap.cpp:109-133: the row index comes from texcoord.u (quirk Q7)
    float theta;
    V2 uv;
    const V3 c = latlong_lookup(T, e.texture, local_dir, theta, uv);
    const float *wr = T.env_tables + e.weight_rows;
    const float row = fminf(fmaxf(uv.u * e.height, 0), e.height - 1);
    const int ri = static_cast<int>(row);
    const float t = row - ri;
    const float denom = fmaxf(fabsf(gl::sinf(theta)), 1e-4f);
    if (t == 0)
        pdf = luminance(c) * wr[ri] * e.normalization / denom;
    else
        pdf = luminance(c) * lerp(wr[ri], wr[ri + 1], t) * e.normalization / denom;
    return c;
}

MCPT_HD float emitter_pdf(const LightTables &T, const EmitterRec &e, V3 look) // emitter.cpp:250-261
{
    if (e.kind == kEmitConstant)
        return k1Div4Pi;
    if (e.kind != kEmitEnvMap)
        return 0;
    float pdf;
    envmap_eval_pdf(T, e, transform_dir(e.to_local, look), pdf);
    return pdf;
}

// Radiance AND sampling pdf of a non-harsh emitter's sampled connection (emitter_eval_sample + emitter_pdf(-wi)).  For the
// environment map the two directions are -M wi and M (-wi): negatives of each other component by component, except that a
// component that is exactly zero may come out with either sign — which changes neither angle (acosf(+-0), atan2f(z, +-0), and
// atan2f(+-0, x) gives +-0 or, wrapped, pi) nor a texel nor a weight (u = +-0: tap 0 with fraction +-0), so one lookup serves both.
MCPT_HD V3 emitter_eval_sample_pdf(const LightTables &T, const EmitterRec &e, const LightSample &s, float &pdf_direct)
{
    if (e.kind == kEmitEnvMap)
        return envmap_eval_pdf(T, e, -transform_dir(e.to_local, s.wi), pdf_direct);
    pdf_direct = emitter_pdf(T, e, -s.wi);
    return emitter_eval_sample(T, e, s);
}

// Radiance along an escaping ray AND the pdf of sampling that direction (emitter_eval_dir + emitter_pdf: the same lookup twice).
MCPT_HD V3 emitter_eval_dir_pdf(const LightTables &T, const EmitterRec &e, V3 look, float &pdf_direct)
{
    if (e.kind == kEmitEnvMap)
        return envmap_eval_pdf(T, e, transform_dir(e.to_local, look), pdf_direct);
    pdf_direct = emitter_pdf(T, e, look);
    return emitter_eval_dir(T, e, look);
}

// ---- homogeneous medium (homogeneous.cpp) -----------------------------------
struct MediumEvent // medium.hpp:55-62
{
    bool valid, scattered;
    float pdf, distance;
    V3 attenuation;
};

MCPT_HD MediumEvent medium_event_init()
{
    MediumEvent m;
    m.valid = false, m.scattered = false, m.pdf = 1.0f, m.distance = 0, m.attenuation = V3{1.0f, 1.0f, 1.0f};
    return m;
}

// exp(-sigma_t d) per channel, the double one (homogeneous.cpp evaluates it per channel and per use).  The values are functions of
// their arguments' bits alone, so they are computed once per distance — and once for all three channels of a gray medium (equal
// coefficients: the three arguments are the same float product), which is what most media are.
struct Exp3
{
    double x, y, z;
};
MCPT_HD Exp3 exp3(V3 sigma_t, float d)
{
    Exp3 e;
    e.x = gl::exp(D(-sigma_t.x * d));
    if (sigma_t.x == sigma_t.y && sigma_t.x == sigma_t.z)
    {
        e.y = e.z = e.x;
        return e;
    }
    e.y = gl::exp(D(-sigma_t.y * d));
    e.z = gl::exp(D(-sigma_t.z * d));
    return e;
}
MCPT_HD V3 to_float3(const Exp3 &e) { return V3{static_cast<float>(e.x), static_cast<float>(e.y), static_cast<float>(e.z)}; }

MCPT_HD V3 transmittance3(V3 sigma_t, float d) { return to_float3(exp3(sigma_t, d)); }

MCPT_HD void medium_sample_distance(const MediumRec &m, float max_distance, uint32_t &rng, MediumEvent &r) // :9-53
{
    const V3 st = from(m.sigma_t);
    float xi0 = lcg_next(rng);
    if (xi0 < m.sampling_weight)
    {
        xi0 /= m.sampling_weight;
        const int channel = static_cast<int>(lcg_next(rng) * 3);
        r.distance = static_cast<float>(-gl::log(D(1.0f - xi0)) / D(comp(st, channel)));
        r.scattered = r.distance < max_distance;
    }
    if (!r.scattered)
        r.distance = max_distance;
    const Exp3 e = exp3(st, r.distance); // (the pdf's terms and the transmittance: the same three values)
    if (r.scattered)
    {
        // accumulated on top of the record's initial pdf of 1
        r.pdf = static_cast<float>(D(r.pdf) + D(st.x) * e.x);
        r.pdf = static_cast<float>(D(r.pdf) + D(st.y) * e.y);
        r.pdf = static_cast<float>(D(r.pdf) + D(st.z) * e.z);
        r.pdf *= m.sampling_weight * (1.0f / 3.0f);
    }
    else
    {
        r.pdf = 0;
        r.pdf = static_cast<float>(D(r.pdf) + e.x);
        r.pdf = static_cast<float>(D(r.pdf) + e.y);
        r.pdf = static_cast<float>(D(r.pdf) + e.z);
        r.pdf = m.sampling_weight * (1.0f / 3.0f) * r.pdf + (1.0f - m.sampling_weight);
    }
    r.attenuation = to_float3(e);
    if (r.attenuation.x > kEpsFloat || r.attenuation.y > kEpsFloat || r.attenuation.z > kEpsFloat)
        r.valid = true;
    if (r.scattered)
        r.attenuation *= from(m.sigma_s);
}

// Transmittance over a known distance (homogeneous.cpp:55-81); r.distance and
// r.scattered are inputs.
MCPT_HD void medium_transmittance(const MediumRec &m, MediumEvent &r)
{
    const V3 st = from(m.sigma_t);
    r.attenuation = transmittance3(st, r.distance);
    if (r.attenuation.x > kEpsFloat || r.attenuation.y > kEpsFloat || r.attenuation.z > kEpsFloat)
        r.valid = true;
    if (!r.valid)
        return;
    if (r.scattered)
    {
        r.pdf += st.x * r.attenuation.x;
        r.pdf += st.y * r.attenuation.y;
        r.pdf += st.z * r.attenuation.z;
        r.pdf *= m.sampling_weight * (1.0f / 3.0f);
        r.attenuation *= from(m.sigma_s);
    }
    else
    {
        r.pdf += r.attenuation.x;
        r.pdf += r.attenuation.y;
        r.pdf += r.attenuation.z;
        r.pdf = m.sampling_weight * (1.0f / 3.0f) * r.pdf + (1.0f - m.sampling_weight);
    }
}

// ---- phase functions (henyey_greenstein.cpp, isotropic.cpp) ------------------
struct PhaseQuery // medium.hpp:27-34
{
    bool valid;
    float pdf;
    V3 wi, wo, attenuation;
};

MCPT_HD void hg_value(V3 g, float cos_t, PhaseQuery &r)
{
    if (g.x == g.y && g.x == g.z)
    {
        // (one asymmetry parameter for the three channels: one division and one square root instead of three, the same bits)
        const float temp = 1.0f + sqr(g.x) + 2.0f * cos_t * g.x;
        const float k = 1.0f / (temp * sqrtf(temp)); // (V3 / V3 multiplies by the reciprocal: vecmath.h)
        r.attenuation = splat((k1Div4Pi * (1.0f - sqr(g.x))) * k);
    }
    else
    {
        const V3 temp = 1.0f + sqr(g) + 2.0f * cos_t * g;
        r.attenuation = k1Div4Pi * (1.0f - sqr(g)) / (temp * vsqrt(temp));
    }
    r.pdf = 0;
    r.pdf += r.attenuation.x;
    r.pdf += r.attenuation.y;
    r.pdf += r.attenuation.z;
    r.pdf *= (1.0f / 3.0f);
}

MCPT_HD void phase_sample(const MediumRec &m, uint32_t &rng, PhaseQuery &r)
{
    r.valid = false;
    if (!m.hg) // isotropic.cpp:9-15
    {
        r.valid = true;
        r.attenuation = splat(k1Div4Pi);
        r.pdf = k1Div4Pi;
        const float xi0 = lcg_next(rng), xi1 = lcg_next(rng);
        r.wi = sample_sphere_uniform(xi0, xi1);
        return;
    }
    // henyey_greenstein.cpp:9-43
    const V3 gv = from(m.g);
    const int channel = static_cast<int>(lcg_next(rng) * 3);
    const float g = comp(gv, channel);
    float cos_t;
    if (fabsf(g) < kEpsFloat)
    {
        cos_t = 1.0f - 2.0f * lcg_next(rng);
    }
    else
    {
        const float term = (1.0f - sqr(g)) / (1.0f - g + 2.0f * g * lcg_next(rng));
        cos_t = (1.0f + sqr(g) - sqr(term)) / (2.0f * g);
    }
    hg_value(gv, cos_t, r);
    if (r.pdf < kEps)
        return;
    r.valid = true;
    const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - sqr(cos_t)));
    const float phi = k2Pi * lcg_next(rng);
    r.wi = -frame_to_world(V3{sin_t * gl::cosf(phi), sin_t * gl::sinf(phi), cos_t}, r.wo);
}

MCPT_HD void phase_eval(const MediumRec &m, PhaseQuery &r)
{
    r.valid = false;
    if (!m.hg) // isotropic.cpp:17-22
    {
        r.valid = true;
        r.attenuation = splat(k1Div4Pi);
        r.pdf = k1Div4Pi;
        return;
    }
    hg_value(from(m.g), dot(-r.wi, r.wo), r); // henyey_greenstein.cpp:45-60
    if (r.pdf < kEps)
        return;
    r.valid = true;
}

} // namespace mcpt

#endif // MCPT_LIGHTS_MEDIA_H
