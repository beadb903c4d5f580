// Reference-side binding of libmcpt_hip.so.
//
// This header is written to be dropped into the REFERENCE tree
// (zhiwei-c/Monte-Carlo-Path-Tracing): it includes the reference's own
// "csrt/renderer/renderer.hpp", serialises a csrt::RendererConfig into the MCSD
// byte layout (include/mcsd_format.h) and drives the C ABI of include/mcpt.h.  A
// maintainer replaces the body of csrt::RayTracer (src/ray_tracer.cpp:124-159) for
// the new backend type with a csrt::HipBackend member:
//
//     HipBackend backend(config);            // commit + upload (renderer.cpp:259-348)
//     backend.Draw(frame_);                  // Renderer::Draw(float*) (renderer.cpp:678-721)
//
// It is not part of libmcpt_hip.so and is compiled in this repository only by
// oracle/ref_driver.cpp (where the reference headers exist) for the round-trip test
// tests/test_oracle_vs_reference.py::test_reference_binding_round_trip: every
// configuration must come back from ToMcsd(...) byte for byte.
#ifndef MCPT_BACKEND_HPP
#define MCPT_BACKEND_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "csrt/renderer/renderer.hpp"

#include "mcpt.h"
#include "mcsd_scene.hpp"

namespace csrt
{

inline void Store3(float *dst, const Vec3 &v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; }

inline void Store16(float *dst, const Mat4 &m)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            dst[4 * r + c] = m[r][c];
}

// csrt::RendererConfig (renderer.hpp:18-28) -> the configuration crossing the C ABI.
inline mcsd::Scene ToMcsd(const RendererConfig &c)
{
    mcsd::Scene s;
    s.camera.spp = c.camera.spp, s.camera.width = c.camera.width, s.camera.height = c.camera.height;
    s.camera.fov_x = c.camera.fov_x;
    Store3(s.camera.eye, c.camera.eye), Store3(s.camera.look_at, c.camera.look_at), Store3(s.camera.up, c.camera.up);

    s.integrator.type = c.integrator.type == IntegratorType::kVolPath ? MCSD_INTEGRATOR_VOLPATH : MCSD_INTEGRATOR_PATH;
    s.integrator.hide_emitters = c.integrator.hide_emitters ? 1u : 0u;
    s.integrator.pdf_rr = c.integrator.pdf_rr;
    s.integrator.depth_rr = c.integrator.depth_rr, s.integrator.depth_max = c.integrator.depth_max;

    for (const TextureInfo &t : c.textures)
    {
        mcsd::Texture o;
        switch (t.type)
        {
        case TextureType::kConstant:
            o.type = MCSD_TEX_CONSTANT;
            Store3(o.color, t.constant.color);
            break;
        case TextureType::kCheckerboard:
            o.type = MCSD_TEX_CHECKERBOARD;
            Store3(o.color0, t.checkerboard.color0), Store3(o.color1, t.checkerboard.color1);
            Store16(o.to_uv, t.checkerboard.to_uv);
            break;
        case TextureType::kBitmap:
            o.type = MCSD_TEX_BITMAP;
            o.width = t.bitmap.width, o.height = t.bitmap.height, o.channel = t.bitmap.channel;
            o.data = t.bitmap.data;
            Store16(o.to_uv, t.bitmap.to_uv);
            break;
        default:
            throw std::runtime_error("unknow texture type.");
        }
        s.textures.push_back(std::move(o));
    }

    for (const BsdfInfo &b : c.bsdfs)
    {
        mcsd::Bsdf o;
        o.twosided = b.twosided ? 1u : 0u;
        o.id_opacity = static_cast<uint32_t>(b.id_opacity), o.id_bump_map = static_cast<uint32_t>(b.id_bump_map);
        switch (b.type)
        {
        case BsdfType::kAreaLight:
            o.type = MCSD_BSDF_AREA_LIGHT;
            o.weight = b.area_light.weight, o.id_radiance = static_cast<uint32_t>(b.area_light.id_radiance);
            break;
        case BsdfType::kDiffuse:
            o.type = MCSD_BSDF_DIFFUSE;
            o.id_diffuse_reflectance = static_cast<uint32_t>(b.diffuse.id_diffuse_reflectance);
            break;
        case BsdfType::kRoughDiffuse:
            o.type = MCSD_BSDF_ROUGH_DIFFUSE;
            o.use_fast_approx = b.rough_diffuse.use_fast_approx ? 1u : 0u;
            o.id_diffuse_reflectance = static_cast<uint32_t>(b.rough_diffuse.id_diffuse_reflectance);
            o.id_roughness = static_cast<uint32_t>(b.rough_diffuse.id_roughness);
            break;
        case BsdfType::kConductor:
            o.type = MCSD_BSDF_CONDUCTOR;
            o.id_roughness_u = static_cast<uint32_t>(b.conductor.id_roughness_u);
            o.id_roughness_v = static_cast<uint32_t>(b.conductor.id_roughness_v);
            o.id_specular_reflectance = static_cast<uint32_t>(b.conductor.id_specular_reflectance);
            Store3(o.reflectivity, b.conductor.reflectivity), Store3(o.edgetint, b.conductor.edgetint);
            break;
        case BsdfType::kDielectric:
        case BsdfType::kThinDielectric:
            o.type = b.type == BsdfType::kDielectric ? MCSD_BSDF_DIELECTRIC : MCSD_BSDF_THIN_DIELECTRIC;
            o.id_roughness_u = static_cast<uint32_t>(b.dielectric.id_roughness_u);
            o.id_roughness_v = static_cast<uint32_t>(b.dielectric.id_roughness_v);
            o.id_specular_reflectance = static_cast<uint32_t>(b.dielectric.id_specular_reflectance);
            o.id_specular_transmittance = static_cast<uint32_t>(b.dielectric.id_specular_transmittance);
            o.eta = b.dielectric.eta;
            break;
        case BsdfType::kPlastic:
            o.type = MCSD_BSDF_PLASTIC;
            o.eta = b.plastic.eta;
            o.id_roughness = static_cast<uint32_t>(b.plastic.id_roughness);
            o.id_diffuse_reflectance = static_cast<uint32_t>(b.plastic.id_diffuse_reflectance);
            o.id_specular_reflectance = static_cast<uint32_t>(b.plastic.id_specular_reflectance);
            break;
        default:
            throw std::runtime_error("unknow BSDF type.");
        }
        s.bsdfs.push_back(o);
    }

    for (const MediumInfo &m : c.media)
    {
        mcsd::Medium o;
        Store3(o.sigma_a, m.homogeneous.sigma_a), Store3(o.sigma_s, m.homogeneous.sigma_s);
        o.phase_type = m.phase_func.type == PhaseFunctionType::kHenyeyGreenstein ? MCSD_PHASE_HG : MCSD_PHASE_ISOTROPIC;
        Store3(o.g, m.phase_func.g);
        s.media.push_back(o);
    }

    for (const InstanceInfo &in : c.instances)
    {
        mcsd::Instance o;
        switch (in.type)
        {
        case InstanceType::kCube: o.type = MCSD_INST_CUBE; break;
        case InstanceType::kRectangle: o.type = MCSD_INST_RECTANGLE; break;
        case InstanceType::kMeshes: o.type = MCSD_INST_MESHES; break;
        case InstanceType::kSphere: o.type = MCSD_INST_SPHERE; break;
        case InstanceType::kDisk: o.type = MCSD_INST_DISK; break;
        case InstanceType::kCylinder: o.type = MCSD_INST_CYLINDER; break;
        default: throw std::runtime_error("unknow instance type.");
        }
        o.id_bsdf = static_cast<uint32_t>(in.id_bsdf);
        o.id_medium_int = static_cast<uint32_t>(in.id_medium_int), o.id_medium_ext = static_cast<uint32_t>(in.id_medium_ext);
        o.flip_normals = in.flip_normals ? 1u : 0u;
        Store16(o.to_world, in.to_world);
        o.sphere_radius = in.sphere.radius;
        Store3(o.sphere_center, in.sphere.center);
        o.cyl_radius = in.cylinder.radius;
        Store3(o.cyl_p0, in.cylinder.p0), Store3(o.cyl_p1, in.cylinder.p1);
        for (const Vec2 &t : in.meshes.texcoords)
            o.texcoords.push_back(t.u), o.texcoords.push_back(t.v);
        auto append3 = [](std::vector<float> &dst, const std::vector<Vec3> &src)
        {
            for (const Vec3 &v : src)
                dst.push_back(v.x), dst.push_back(v.y), dst.push_back(v.z);
        };
        append3(o.positions, in.meshes.positions), append3(o.normals, in.meshes.normals);
        append3(o.tangents, in.meshes.tangents), append3(o.bitangents, in.meshes.bitangents);
        for (const Uvec3 &i : in.meshes.indices)
            o.indices.push_back(i.x), o.indices.push_back(i.y), o.indices.push_back(i.z);
        s.instances.push_back(std::move(o));
    }

    for (const EmitterInfo &e : c.emitters)
    {
        mcsd::Emitter o;
        switch (e.type)
        {
        case EmitterType::kPoint:
            o.type = MCSD_EMIT_POINT;
            Store3(o.position, e.point.position), Store3(o.intensity, e.point.intensity);
            break;
        case EmitterType::kSpot:
            o.type = MCSD_EMIT_SPOT;
            o.cutoff_angle = e.spot.cutoff_angle, o.beam_width = e.spot.beam_width;
            o.id_texture = static_cast<uint32_t>(e.spot.id_texture);
            Store3(o.intensity, e.spot.intensity);
            Store16(o.to_world, e.spot.to_world);
            break;
        case EmitterType::kDirectional:
            o.type = MCSD_EMIT_DIRECTIONAL;
            Store3(o.direction, e.directional.direction), Store3(o.radiance, e.directional.radiance);
            break;
        case EmitterType::kSun:
            o.type = MCSD_EMIT_SUN;
            o.cos_cutoff_angle = e.sun.cos_cutoff_angle, o.id_texture = static_cast<uint32_t>(e.sun.id_texture);
            Store3(o.direction, e.sun.direction), Store3(o.radiance, e.sun.radiance);
            break;
        case EmitterType::kEnvMap:
            o.type = MCSD_EMIT_ENVMAP;
            o.id_radiance = static_cast<uint32_t>(e.envmap.id_radiance);
            Store16(o.to_world, e.envmap.to_world);
            break;
        case EmitterType::kConstant:
            o.type = MCSD_EMIT_CONSTANT;
            Store3(o.radiance, e.constant.radiance);
            break;
        default:
            throw std::runtime_error("unknow emitter type.");
        }
        s.emitters.push_back(o);
    }
    return s;
}

#if !defined(MCPT_BACKEND_NO_LIBRARY)
// The renderer behind the seam: same life cycle as csrt::Renderer.
class HipBackend
{
public:
    explicit HipBackend(const RendererConfig &config, int device = 0)
    {
        const std::vector<uint8_t> bytes = mcsd::Serialize(ToMcsd(config));
        mcpt_config *cfg = nullptr;
        if (mcpt_config_from_mcsd_bytes(bytes.data(), bytes.size(), &cfg) != 0)
            throw MyException(mcpt_last_error());
        const int rc = mcpt_renderer_create(cfg, device, &renderer_);
        mcpt_config_destroy(cfg);
        if (rc != 0)
            throw MyException(mcpt_last_error()); // "error when commit renderer.\n\t..." like renderer.cpp:342-347
    }
    HipBackend(const HipBackend &) = delete;
    HipBackend &operator=(const HipBackend &) = delete;
    ~HipBackend() { mcpt_renderer_destroy(renderer_); }

    // Renderer::Draw(float *frame) const: blocking, caller-owned host frame (width*height*3 floats)
    void Draw(float *frame) const
    {
        if (mcpt_renderer_draw(renderer_, frame, nullptr) != 0)
            throw MyException(std::string("error when draw.\n\t") + mcpt_last_error());
    }

private:
    mcpt_renderer *renderer_ = nullptr;
};
#endif

} // namespace csrt

#endif // MCPT_BACKEND_HPP
