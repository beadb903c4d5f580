// TEST INFRASTRUCTURE — CPU oracle, part 5: the two integrators.
//
// Restates reference src/renderer/integrators/path.cpp (ShadePath :8-136,
// direct light :138-236, BSDF record helpers :238-296) and volpath.cpp
// (ShadeVolPath :8-245, direct light at a surface vertex :247-375 and at a
// medium vertex :377-485).  One LCG state per pixel flows through every call;
// multi-draw call sites take their numbers explicitly left to right.
#ifndef ORACLE_INTEGRATORS_HPP
#define ORACLE_INTEGRATORS_HPP

#include "oracle_trace.hpp"

namespace orc
{

inline Scatter RecordFromHit(const Hit &hit, V3 wo)
{
    Scatter r;
    r.wo = wo;
    r.uv = hit.uv;
    r.position = hit.position;
    return r;
}

inline void OrientRecord(const Hit &hit, V3 facing, Scatter *r)
{
    r->inside = hit.inside;
    r->normal = hit.normal;
    r->tangent = hit.tangent;
    r->bitangent = hit.bitangent;
    if (Dot(facing, hit.normal) < 0.0f)
    {
        r->inside = !r->inside;
        r->normal = -r->normal;
    }
}

// path.cpp:238-266
inline Scatter EvalAtHit(const Scene &sc, V3 wi, V3 wo, const Hit &hit, const Bsdf *bsdf)
{
    Scatter r = RecordFromHit(hit, wo);
    r.wi = wi;
    if (bsdf)
    {
        OrientRecord(hit, -wi, &r);
        BsdfEval(sc, *bsdf, &r);
    }
    else
    {
        r.pdf = 1;
        r.attenuation = V3(1);
        r.valid = true;
    }
    return r;
}

// path.cpp:268-296: a shape without BSDF lets the ray continue straight on
inline Scatter SampleAtHit(const Scene &sc, V3 wo, const Hit &hit, const Bsdf *bsdf, uint32_t *rng)
{
    Scatter r = RecordFromHit(hit, wo);
    if (bsdf != nullptr)
    {
        OrientRecord(hit, wo, &r);
        BsdfSample(sc, *bsdf, rng, &r);
    }
    else
    {
        r.wi = wo;
        r.pdf = 1.0f;
        r.attenuation = V3(1.0f);
        r.valid = true;
    }
    return r;
}

// Shared by all three direct-light routines: pick the area light, sample a
// point on it, trace the shadow ray FROM the light TOWARDS the shading point
// (path.cpp:185-205).  Returns false when occluded or back-facing.
struct AreaConnection
{
    uint32_t light = 0, inst = 0;
    Hit on_light;
    V3 wi;
    float distance = 0, cos_light = 0;
};

inline bool ConnectAreaLight(const Scene &sc, V3 shade_pos, uint32_t *rng,
                             AreaConnection *c, TraceStats *st)
{
    const float xi_pick = Rand(rng);
    c->light = CdfSearch(static_cast<uint32_t>(sc.light_cdf.size()), sc.light_cdf.data(), xi_pick) - 1;
    c->inst = sc.light_inst[c->light];
    const float xi0 = Rand(rng), xi1 = Rand(rng), xi2 = Rand(rng);
    c->on_light = SampleInstance(sc, c->inst, xi0, xi1, xi2);
    const V3 d = shade_pos - c->on_light.position;
    c->distance = Len(d);
    Ray shadow = MakeRay(c->on_light.position, Unit(d));
    shadow.t_max = c->distance - kEpsDistance;
    if (AnyHit(sc, rng, &shadow, st))
        return false;
    c->wi = Unit(d);
    c->cos_light = Dot(c->wi, c->on_light.normal);
    return c->cos_light >= kEpsFloat;
}

inline float AreaLightPdf(const Scene &sc, uint32_t light, uint32_t inst, float distance, float cos_light)
{
    const float pdf_area = (sc.light_cdf[light + 1] - sc.light_cdf[light]) * sc.inst_pdf_area[inst];
    return pdf_area * Sq(distance) / cos_light;
}

// path.cpp:138-236
inline V3 DirectLightPath(const Scene &sc, const Hit &hit, V3 wo, uint32_t *rng, TraceStats *st)
{
    V3 L(0);
    const Bsdf *bsdf = InstanceBsdf(sc, hit.inst);
    for (const Emitter &e : sc.emitters)
    {
        const float xi0 = Rand(rng), xi1 = Rand(rng);
        const LightSample s = EmitterSample(sc, e, hit.position, xi0, xi1);
        Ray shadow = MakeRay(hit.position, -s.wi);
        shadow.t_max = s.distance - kEpsDistance;
        if (AnyHit(sc, rng, &shadow, st))
            continue;
        if (Dot(-s.wi, hit.normal) < kEpsFloat)
            continue;
        const Scatter r = EvalAtHit(sc, s.wi, wo, hit, bsdf);
        if (!r.valid)
            continue;
        const V3 radiance = EmitterEvalSample(sc, e, s);
        if (s.harsh)
        {
            L += radiance * r.attenuation;
        }
        else
        {
            const float pdf_direct = EmitterPdf(sc, e, -s.wi);
            if (pdf_direct > kEpsFloat)
            {
                const float w = PowerHeuristic(pdf_direct, r.pdf);
                L += w * radiance * (r.attenuation / pdf_direct);
            }
        }
    }
    if (!sc.light_inst.empty())
    {
        AreaConnection c;
        if (!ConnectAreaLight(sc, hit.position, rng, &c, st))
            return L;
        if (Dot(-c.wi, hit.normal) < kEpsFloat)
            return L;
        const Scatter r = EvalAtHit(sc, c.wi, wo, hit, bsdf);
        if (!r.valid)
            return L;
        const float pdf_direct = AreaLightPdf(sc, c.light, c.inst, c.distance, c.cos_light),
                    w = PowerHeuristic(pdf_direct, r.pdf);
        const V3 radiance = BsdfRadiance(sc, sc.bsdfs[sc.inst_bsdf[c.inst]], c.on_light.uv);
        L += w * radiance * (r.attenuation / pdf_direct);
    }
    return L;
}

inline V3 EscapedPrimary(const Scene &sc, V3 look)
{
    V3 L(0);
    if (sc.id_envmap != kNone)
        L += EmitterEvalDir(sc, sc.emitters[sc.id_envmap], look);
    if (sc.id_sun != kNone)
        L += EmitterEvalDir(sc, sc.emitters[sc.id_sun], look);
    return L;
}

// path.cpp:8-136
inline V3 ShadePath(const Scene &sc, V3 eye, V3 look, uint32_t *rng, TraceStats *st = nullptr)
{
    V3 L(0);
    Ray ray = MakeRay(eye, look);
    Hit hit = ClosestHit(sc, rng, &ray, st);
    if (!hit.valid)
        return EscapedPrimary(sc, look);
    const Bsdf *bsdf = InstanceBsdf(sc, hit.inst);
    if (bsdf != nullptr)
    {
        if (hit.inside && !bsdf->twosided)
            return V3(0);
        if (bsdf->type == Mat::kAreaLight)
            return sc.hide_emitters ? V3(0) : BsdfRadiance(sc, *bsdf, hit.uv);
    }
    V3 throughput(1), wo = -look;
    for (uint32_t depth = 1;
         depth < sc.depth_rr || (depth < sc.depth_max && Rand(rng) < sc.pdf_rr);
         ++depth)
    {
        L += throughput * DirectLightPath(sc, hit, wo, rng, st);
        const Scatter r = SampleAtHit(sc, wo, hit, bsdf, rng);
        if (!r.valid)
            break;
        throughput *= r.attenuation / r.pdf;
        if (MaxComp(throughput) < kEps)
            break;
        ray = MakeRay(r.position, -r.wi);
        hit = ClosestHit(sc, rng, &ray, st);
        if (!hit.valid)
        {
            if (sc.id_envmap != kNone)
            {
                const Emitter &env = sc.emitters[sc.id_envmap];
                const V3 radiance = EmitterEvalDir(sc, env, -r.wi);
                const float pdf_direct = EmitterPdf(sc, env, -r.wi),
                            w = PowerHeuristic(r.pdf, pdf_direct);
                L += w * throughput * radiance;
            }
            break;
        }
        bsdf = InstanceBsdf(sc, hit.inst);
        if (bsdf != nullptr)
        {
            if (hit.inside && !bsdf->twosided)
                break;
            if (bsdf->type == Mat::kAreaLight)
            {
                const float cos_light = Dot(r.wi, hit.normal);
                if (cos_light < kEpsFloat)
                    break;
                const float pdf_direct = AreaLightPdf(sc, sc.inst_light[hit.inst], hit.inst, ray.t_max, cos_light),
                            w = PowerHeuristic(r.pdf, pdf_direct);
                const V3 radiance = BsdfRadiance(sc, *bsdf, hit.uv);
                L += w * throughput * radiance;
                break;
            }
        }
        wo = r.wi;
        if (depth >= sc.depth_rr)
            throughput *= sc.rr_scale; // multiplies by pdf_rr (quirk Q2)
    }
    return L;
}

// ---------------------------------------------------------------------------
// Volumetric path tracer
// ---------------------------------------------------------------------------
inline uint32_t MediumOnSide(const Hit &hit, V3 w) // volpath.cpp:44-45,163-166,253-254
{
    const bool inside = Dot(w, hit.normal) > 0 ? hit.inside : !hit.inside;
    return inside ? hit.med_int : hit.med_ext;
}

// volpath.cpp:247-375
inline V3 DirectLightVolSurface(const Scene &sc, const Hit &hit, V3 wo, uint32_t *rng, TraceStats *st)
{
    V3 L(0);
    const uint32_t id_medium = MediumOnSide(hit, wo);
    const Medium *medium = id_medium != kNone ? &sc.media[id_medium] : nullptr;
    const Bsdf *bsdf = InstanceBsdf(sc, hit.inst);
    for (const Emitter &e : sc.emitters)
    {
        const float xi0 = Rand(rng), xi1 = Rand(rng);
        const LightSample s = EmitterSample(sc, e, hit.position, xi0, xi1);
        Ray shadow = MakeRay(hit.position, -s.wi);
        shadow.t_max = s.distance - kEpsDistance;
        if (AnyHit(sc, rng, &shadow, st))
            continue;
        if (Dot(-s.wi, hit.normal) < kEpsFloat)
            continue;
        V3 tr = V3(1.0f);
        if (medium != nullptr)
        {
            MediumSample m;
            m.distance = s.distance;
            MediumTransmittance(*medium, &m);
            if (!m.valid)
                continue;
            tr = m.attenuation / m.pdf;
        }
        const Scatter r = EvalAtHit(sc, s.wi, wo, hit, bsdf);
        if (!r.valid)
            continue;
        const V3 radiance = EmitterEvalSample(sc, e, s);
        if (s.harsh)
        {
            L += radiance * tr * r.attenuation;
        }
        else
        {
            const float pdf_direct = EmitterPdf(sc, e, -s.wi);
            if (pdf_direct > kEpsFloat)
            {
                const float w = PowerHeuristic(pdf_direct, r.pdf);
                L += w * radiance * tr * r.attenuation / pdf_direct;
            }
        }
    }
    if (!sc.light_inst.empty())
    {
        AreaConnection c;
        if (!ConnectAreaLight(sc, hit.position, rng, &c, st))
            return L;
        if (Dot(-c.wi, hit.normal) < kEpsFloat)
            return L;
        V3 tr = V3(1.0f);
        if (medium != nullptr)
        {
            MediumSample m;
            m.distance = c.distance;
            MediumTransmittance(*medium, &m);
            if (!m.valid)
                return L;
            tr = m.attenuation / m.pdf;
        }
        const Scatter r = EvalAtHit(sc, c.wi, wo, hit, bsdf);
        if (!r.valid)
            return L;
        const float pdf_direct = AreaLightPdf(sc, c.light, c.inst, c.distance, c.cos_light),
                    w = PowerHeuristic(pdf_direct, r.pdf);
        const V3 radiance = BsdfRadiance(sc, sc.bsdfs[sc.inst_bsdf[c.inst]], c.on_light.uv);
        L += w * (radiance * tr * r.attenuation / pdf_direct);
    }
    return L;
}

// volpath.cpp:377-485
inline V3 DirectLightVolMedium(const Scene &sc, V3 position, const Medium &medium,
                               V3 wo, uint32_t *rng, TraceStats *st)
{
    V3 L(0);
    for (const Emitter &e : sc.emitters)
    {
        const float xi0 = Rand(rng), xi1 = Rand(rng);
        const LightSample s = EmitterSample(sc, e, position, xi0, xi1);
        Ray shadow = MakeRay(position, -s.wi);
        shadow.t_max = s.distance - kEpsDistance;
        if (AnyHit(sc, rng, &shadow, st))
            continue;
        MediumSample m;
        m.distance = s.distance;
        MediumTransmittance(medium, &m);
        if (!m.valid)
            continue;
        const V3 tr = m.attenuation / m.pdf;
        PhaseSample p;
        p.wi = s.wi, p.wo = wo;
        PhaseEval(medium, &p);
        if (!p.valid)
            continue;
        const V3 radiance = EmitterEvalSample(sc, e, s);
        if (s.harsh)
        {
            L += radiance * tr * p.attenuation;
        }
        else
        {
            const float pdf_direct = EmitterPdf(sc, e, -s.wi);
            if (pdf_direct > kEpsFloat)
            {
                const float w = PowerHeuristic(pdf_direct, p.pdf);
                L += w * radiance * tr * p.attenuation / pdf_direct;
            }
        }
    }
    if (!sc.light_inst.empty())
    {
        AreaConnection c;
        if (!ConnectAreaLight(sc, position, rng, &c, st))
            return L;
        MediumSample m;
        m.distance = c.distance;
        MediumTransmittance(medium, &m);
        if (!m.valid)
            return L;
        const V3 tr = m.attenuation / m.pdf;
        PhaseSample p;
        p.wi = c.wi, p.wo = wo;
        PhaseEval(medium, &p);
        if (!p.valid)
            return L;
        const float pdf_direct = AreaLightPdf(sc, c.light, c.inst, c.distance, c.cos_light),
                    w = PowerHeuristic(pdf_direct, p.pdf);
        const V3 radiance = BsdfRadiance(sc, sc.bsdfs[sc.inst_bsdf[c.inst]], c.on_light.uv);
        L += w * (radiance * tr * p.attenuation / pdf_direct);
    }
    return L;
}

// volpath.cpp:8-245
inline V3 ShadeVolPath(const Scene &sc, V3 eye, V3 look, uint32_t *rng, TraceStats *st = nullptr)
{
    V3 L(0);
    Ray ray = MakeRay(eye, look);
    Hit hit = ClosestHit(sc, rng, &ray, st);
    if (!hit.valid)
        return EscapedPrimary(sc, look);

    V3 throughput(1), wo = -look;
    bool in_medium = false;
    V3 medium_pos;
    const Medium *cur_medium = nullptr;

    // free-flight sampling between the eye and the first surface
    {
        const uint32_t id = MediumOnSide(hit, wo);
        if (id != kNone)
        {
            const Medium *medium = &sc.media[id];
            MediumSample m;
            MediumDistance(*medium, ray.t_max, rng, &m);
            if (m.valid)
            {
                throughput *= m.attenuation / m.pdf;
                if (m.scattered)
                {
                    in_medium = true;
                    medium_pos = ray.origin + ray.dir * m.distance;
                    cur_medium = medium;
                }
            }
        }
    }

    const Bsdf *bsdf = nullptr;
    if (!in_medium)
    {
        bsdf = InstanceBsdf(sc, hit.inst);
        if (bsdf != nullptr)
        {
            if (hit.inside && !bsdf->twosided)
                return V3(0);
            if (bsdf->type == Mat::kAreaLight)
                return sc.hide_emitters ? V3(0) : BsdfRadiance(sc, *bsdf, hit.uv);
        }
    }

    V3 wi;
    float pdf_sample = 0;
    for (uint32_t depth = 1;
         depth < sc.depth_rr || (depth < sc.depth_max && Rand(rng) < sc.pdf_rr);
         ++depth)
    {
        if (in_medium)
        {
            L += throughput * DirectLightVolMedium(sc, medium_pos, *cur_medium, wo, rng, st);
            PhaseSample p;
            p.wo = wo;
            PhaseSampleDir(*cur_medium, rng, &p);
            if (!p.valid)
                break;
            wi = p.wi;
            throughput *= p.attenuation / p.pdf;
            pdf_sample = p.pdf;
            if (MaxComp(throughput) < kEps)
                break;
            ray = MakeRay(medium_pos, -wi);
            hit = ClosestHit(sc, rng, &ray, st);
            MediumSample m;
            MediumDistance(*cur_medium, ray.t_max, rng, &m);
            if (m.valid)
            {
                throughput *= m.attenuation / m.pdf;
                if (m.scattered)
                {
                    in_medium = true;
                    medium_pos = ray.origin + ray.dir * m.distance;
                }
                else
                {
                    in_medium = false;
                }
            }
            else
            {
                in_medium = false;
            }
        }
        else
        {
            L += throughput * DirectLightVolSurface(sc, hit, wo, rng, st);
            const Scatter r = SampleAtHit(sc, wo, hit, bsdf, rng);
            if (!r.valid)
                break;
            wi = r.wi;
            pdf_sample = r.pdf;
            throughput *= r.attenuation / pdf_sample;
            if (MaxComp(throughput) < kEps)
                break;
            ray = MakeRay(r.position, -wi);
            hit = ClosestHit(sc, rng, &ray, st);
            const uint32_t id = MediumOnSide(hit, wi);
            if (id != kNone)
            {
                const Medium *medium = &sc.media[id];
                MediumSample m;
                MediumDistance(*medium, ray.t_max, rng, &m);
                if (m.valid)
                {
                    throughput *= m.attenuation / m.pdf;
                    if (m.scattered)
                    {
                        in_medium = true;
                        medium_pos = ray.origin + ray.dir * m.distance;
                        cur_medium = medium;
                    }
                }
            }
        }

        if (!in_medium)
        {
            if (!hit.valid)
            {
                if (sc.id_envmap != kNone)
                {
                    const Emitter &env = sc.emitters[sc.id_envmap];
                    const V3 radiance = EmitterEvalDir(sc, env, -wi);
                    const float pdf_direct = EmitterPdf(sc, env, -wi),
                                w = PowerHeuristic(pdf_sample, pdf_direct);
                    L += w * throughput * radiance;
                }
                break;
            }
            bsdf = InstanceBsdf(sc, hit.inst);
            if (bsdf != nullptr)
            {
                if (hit.inside && !bsdf->twosided)
                    break;
                if (bsdf->type == Mat::kAreaLight)
                {
                    const float cos_light = Dot(wi, hit.normal);
                    if (cos_light < kEpsFloat)
                        break;
                    const float pdf_direct = AreaLightPdf(sc, sc.inst_light[hit.inst], hit.inst, ray.t_max, cos_light),
                                w = PowerHeuristic(pdf_sample, pdf_direct);
                    const V3 radiance = BsdfRadiance(sc, *bsdf, hit.uv);
                    L += w * throughput * radiance;
                    break;
                }
            }
            wo = wi;
            if (depth >= sc.depth_rr)
                throughput *= sc.rr_scale;
        }
    }
    return L;
}

} // namespace orc

#endif // ORACLE_INTEGRATORS_HPP
