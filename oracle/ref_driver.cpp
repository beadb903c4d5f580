// TEST INFRASTRUCTURE — not part of the product.
//
// Driver for the *compiled reference*: it is linked (by oracle/Makefile, in
// this container only) against the reference's own, unmodified sources under
// /root/reference/src/{renderer,rtcore,tensor,utils} and exposes a few C
// entry points so that the tests can run the real reference on an MCSD scene
// (see include/mcsd_format.h) and pin the oracle restatement against it.
//
// The reference's XML/mesh/image front end (pugixml, assimp, tinyexr) cannot
// be built here, so the driver fills `csrt::RendererConfig`
// (reference include/csrt/renderer/renderer.hpp:18-28) directly from the
// MCSD records and then does exactly what `csrt::RayTracer` does
// (reference src/ray_tracer.cpp:124-159): construct `csrt::Renderer`, call
// `Renderer::Draw(float*)`.
//
// Nothing in this file restates reference code; it only calls it.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "csrt/renderer/bsdfs/kulla_conty.hpp"
#include "csrt/renderer/bsdfs/microfacet.hpp"
#include "csrt/renderer/renderer.hpp"
#include "csrt/rtcore/accel/bvh_builder.hpp"

#include "mcsd_scene.hpp"

namespace
{

thread_local std::string g_error;

csrt::Vec3 V3(const float *v) { return csrt::Vec3{v[0], v[1], v[2]}; }

csrt::Mat4 M4(const float *m)
{
    return csrt::Mat4{m[0], m[1], m[2],  m[3],  m[4],  m[5],  m[6],  m[7],
                      m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]};
}

csrt::RendererConfig ToConfig(const mcsd::Scene &s)
{
    csrt::RendererConfig cfg;
    cfg.backend_type = csrt::BackendType::kCpu;
    cfg.camera.spp = s.camera.spp;
    cfg.camera.width = s.camera.width;
    cfg.camera.height = s.camera.height;
    cfg.camera.fov_x = s.camera.fov_x;
    cfg.camera.eye = V3(s.camera.eye);
    cfg.camera.look_at = V3(s.camera.look_at);
    cfg.camera.up = V3(s.camera.up);

    cfg.integrator.type = s.integrator.type == MCSD_INTEGRATOR_VOLPATH
                              ? csrt::IntegratorType::kVolPath
                              : csrt::IntegratorType::kPath;
    cfg.integrator.hide_emitters = s.integrator.hide_emitters != 0;
    cfg.integrator.pdf_rr = s.integrator.pdf_rr;
    cfg.integrator.depth_rr = s.integrator.depth_rr;
    cfg.integrator.depth_max = s.integrator.depth_max;

    for (const mcsd::Texture &t : s.textures)
    {
        csrt::TextureInfo info;
        switch (t.type)
        {
        case MCSD_TEX_CONSTANT:
            info.type = csrt::TextureType::kConstant;
            info.constant.color = V3(t.color);
            break;
        case MCSD_TEX_CHECKERBOARD:
            info.type = csrt::TextureType::kCheckerboard;
            info.checkerboard.color0 = V3(t.color0);
            info.checkerboard.color1 = V3(t.color1);
            info.checkerboard.to_uv = M4(t.to_uv);
            break;
        case MCSD_TEX_BITMAP:
            info.type = csrt::TextureType::kBitmap;
            info.bitmap.width = t.width;
            info.bitmap.height = t.height;
            info.bitmap.channel = t.channel;
            info.bitmap.data = t.data;
            info.bitmap.to_uv = M4(t.to_uv);
            break;
        }
        cfg.textures.push_back(info);
    }

    for (const mcsd::Bsdf &b : s.bsdfs)
    {
        csrt::BsdfInfo info;
        info.twosided = b.twosided != 0;
        info.id_opacity = b.id_opacity;
        info.id_bump_map = b.id_bump_map;
        switch (b.type)
        {
        case MCSD_BSDF_AREA_LIGHT:
            info.type = csrt::BsdfType::kAreaLight;
            info.area_light.weight = b.weight;
            info.area_light.id_radiance = b.id_radiance;
            break;
        case MCSD_BSDF_DIFFUSE:
            info.type = csrt::BsdfType::kDiffuse;
            info.diffuse.id_diffuse_reflectance = b.id_diffuse_reflectance;
            break;
        case MCSD_BSDF_ROUGH_DIFFUSE:
            info.type = csrt::BsdfType::kRoughDiffuse;
            info.rough_diffuse.use_fast_approx = b.use_fast_approx != 0;
            info.rough_diffuse.id_diffuse_reflectance =
                b.id_diffuse_reflectance;
            info.rough_diffuse.id_roughness = b.id_roughness;
            break;
        case MCSD_BSDF_CONDUCTOR:
            info.type = csrt::BsdfType::kConductor;
            info.conductor.id_roughness_u = b.id_roughness_u;
            info.conductor.id_roughness_v = b.id_roughness_v;
            info.conductor.id_specular_reflectance = b.id_specular_reflectance;
            info.conductor.reflectivity = V3(b.reflectivity);
            info.conductor.edgetint = V3(b.edgetint);
            break;
        case MCSD_BSDF_DIELECTRIC:
        case MCSD_BSDF_THIN_DIELECTRIC:
            info.type = b.type == MCSD_BSDF_DIELECTRIC
                            ? csrt::BsdfType::kDielectric
                            : csrt::BsdfType::kThinDielectric;
            info.dielectric.id_roughness_u = b.id_roughness_u;
            info.dielectric.id_roughness_v = b.id_roughness_v;
            info.dielectric.id_specular_reflectance = b.id_specular_reflectance;
            info.dielectric.id_specular_transmittance =
                b.id_specular_transmittance;
            info.dielectric.eta = b.eta;
            break;
        case MCSD_BSDF_PLASTIC:
            info.type = csrt::BsdfType::kPlastic;
            info.plastic.eta = b.eta;
            info.plastic.id_roughness = b.id_roughness;
            info.plastic.id_diffuse_reflectance = b.id_diffuse_reflectance;
            info.plastic.id_specular_reflectance = b.id_specular_reflectance;
            break;
        }
        cfg.bsdfs.push_back(info);
    }

    for (const mcsd::Medium &m : s.media)
    {
        csrt::MediumInfo info;
        info.type = csrt::MediumType::kHomogeneous;
        info.homogeneous.sigma_a = V3(m.sigma_a);
        info.homogeneous.sigma_s = V3(m.sigma_s);
        info.phase_func.type = m.phase_type == MCSD_PHASE_HG
                                   ? csrt::PhaseFunctionType::kHenyeyGreenstein
                                   : csrt::PhaseFunctionType::kIsotropic;
        info.phase_func.g = V3(m.g);
        cfg.media.push_back(info);
    }

    for (const mcsd::Instance &in : s.instances)
    {
        csrt::InstanceInfo info;
        switch (in.type)
        {
        case MCSD_INST_CUBE:
            info.type = csrt::InstanceType::kCube;
            break;
        case MCSD_INST_RECTANGLE:
            info.type = csrt::InstanceType::kRectangle;
            break;
        case MCSD_INST_MESHES:
            info.type = csrt::InstanceType::kMeshes;
            break;
        case MCSD_INST_SPHERE:
            info.type = csrt::InstanceType::kSphere;
            break;
        case MCSD_INST_DISK:
            info.type = csrt::InstanceType::kDisk;
            break;
        case MCSD_INST_CYLINDER:
            info.type = csrt::InstanceType::kCylinder;
            break;
        }
        info.id_bsdf = in.id_bsdf;
        info.id_medium_int = in.id_medium_int;
        info.id_medium_ext = in.id_medium_ext;
        info.flip_normals = in.flip_normals != 0;
        info.to_world = M4(in.to_world);
        info.sphere.radius = in.sphere_radius;
        info.sphere.center = V3(in.sphere_center);
        info.cylinder.radius = in.cyl_radius;
        info.cylinder.p0 = V3(in.cyl_p0);
        info.cylinder.p1 = V3(in.cyl_p1);
        for (size_t k = 0; k + 1 < in.texcoords.size(); k += 2)
            info.meshes.texcoords.push_back(
                csrt::Vec2{in.texcoords[k], in.texcoords[k + 1]});
        for (size_t k = 0; k + 2 < in.positions.size(); k += 3)
            info.meshes.positions.push_back(V3(&in.positions[k]));
        for (size_t k = 0; k + 2 < in.normals.size(); k += 3)
            info.meshes.normals.push_back(V3(&in.normals[k]));
        for (size_t k = 0; k + 2 < in.tangents.size(); k += 3)
            info.meshes.tangents.push_back(V3(&in.tangents[k]));
        for (size_t k = 0; k + 2 < in.bitangents.size(); k += 3)
            info.meshes.bitangents.push_back(V3(&in.bitangents[k]));
        for (size_t k = 0; k + 2 < in.indices.size(); k += 3)
            info.meshes.indices.push_back(csrt::Uvec3{
                in.indices[k], in.indices[k + 1], in.indices[k + 2]});
        cfg.instances.push_back(info);
    }

    for (const mcsd::Emitter &e : s.emitters)
    {
        csrt::EmitterInfo info;
        switch (e.type)
        {
        case MCSD_EMIT_POINT:
            info.type = csrt::EmitterType::kPoint;
            info.point.position = V3(e.position);
            info.point.intensity = V3(e.intensity);
            break;
        case MCSD_EMIT_SPOT:
            info.type = csrt::EmitterType::kSpot;
            info.spot = csrt::SpotLightInfo{};
            info.spot.cutoff_angle = e.cutoff_angle;
            info.spot.beam_width = e.beam_width;
            info.spot.id_texture = e.id_texture;
            info.spot.intensity = V3(e.intensity);
            info.spot.to_world = M4(e.to_world);
            break;
        case MCSD_EMIT_DIRECTIONAL:
            info.type = csrt::EmitterType::kDirectional;
            info.directional.direction = V3(e.direction);
            info.directional.radiance = V3(e.radiance);
            break;
        case MCSD_EMIT_SUN:
            info.type = csrt::EmitterType::kSun;
            info.sun = csrt::SunInfo{};
            info.sun.cos_cutoff_angle = e.cos_cutoff_angle;
            info.sun.id_texture = e.id_texture;
            info.sun.direction = V3(e.direction);
            info.sun.radiance = V3(e.radiance);
            break;
        case MCSD_EMIT_ENVMAP:
            info.type = csrt::EmitterType::kEnvMap;
            info.envmap = csrt::EnvMapInfo{};
            info.envmap.id_radiance = e.id_radiance;
            info.envmap.to_world = M4(e.to_world);
            break;
        case MCSD_EMIT_CONSTANT:
            info.type = csrt::EmitterType::kConstant;
            info.constant.radiance = V3(e.radiance);
            break;
        }
        cfg.emitters.push_back(info);
    }
    return cfg;
}

} // namespace

extern "C"
{

const char *mcpt_ref_last_error(void) { return g_error.c_str(); }

// Renders `mcsd_path` with the compiled reference's CPU backend into `frame`
// (width*height*3 float32, row 0 = top).  `render_seconds` receives the wall
// time of Renderer::Draw only (the reference's own Timer brackets the same
// region, renderer.cpp:144,252).  Returns 0 on success.
int mcpt_ref_render(const char *mcsd_path, float *frame, double *render_seconds)
{
    try
    {
        const mcsd::Scene scene = mcsd::Load(mcsd_path);
        const csrt::RendererConfig cfg = ToConfig(scene);
        csrt::Renderer renderer(cfg);
        const auto t0 = std::chrono::steady_clock::now();
        renderer.Draw(frame);
        const auto t1 = std::chrono::steady_clock::now();
        if (render_seconds)
            *render_seconds = std::chrono::duration<double>(t1 - t0).count();
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// ---- known-answer hooks: thin calls into reference functions -------------

uint32_t mcpt_ref_tea4(uint32_t v0, uint32_t v1) { return csrt::Tea<4>(v0, v1); }

float mcpt_ref_random_float(uint32_t *seed) { return csrt::RandomFloat(seed); }

float mcpt_ref_vdc2(uint32_t i) { return csrt::GetVanDerCorputSequence<2>(i); }

float mcpt_ref_vdc3(uint32_t i) { return csrt::GetVanDerCorputSequence<3>(i); }

// Kulla-Conty tables (kulla_conty.cpp:62-80): brdf[128*128], albedo[128].
void mcpt_ref_kulla_conty(float *brdf, float *albedo)
{
    csrt::ComputeKullaConty(brdf, albedo);
}

// LBVH build (bvh_builder.cpp:74-207).  aabbs: n*6 floats (min xyz, max xyz);
// outputs 2n-1 nodes: leaf/left/right/object (u32 each), area, aabb[6].
int mcpt_ref_bvh_build(uint32_t n, const float *aabbs, const float *areas,
                       uint32_t *leaf, uint32_t *left, uint32_t *right,
                       uint32_t *object, float *area, float *box)
{
    try
    {
        std::vector<csrt::AABB> in(n);
        std::vector<float> ar(areas, areas + n);
        for (uint32_t i = 0; i < n; ++i)
            in[i] = csrt::AABB(V3(aabbs + 6 * i), V3(aabbs + 6 * i + 3));
        const std::vector<csrt::BvhNode> nodes = csrt::BvhBuilder::Build(in, ar);
        for (size_t i = 0; i < nodes.size(); ++i)
        {
            leaf[i] = nodes[i].leaf ? 1u : 0u;
            left[i] = nodes[i].id_left;
            right[i] = nodes[i].id_right;
            object[i] = nodes[i].id_object;
            area[i] = nodes[i].area;
            const csrt::Vec3 lo = nodes[i].aabb.min(), hi = nodes[i].aabb.max();
            box[6 * i + 0] = lo.x, box[6 * i + 1] = lo.y, box[6 * i + 2] = lo.z;
            box[6 * i + 3] = hi.x, box[6 * i + 4] = hi.y, box[6 * i + 5] = hi.z;
        }
        return static_cast<int>(nodes.size());
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return -1;
    }
}

} // extern "C"

#ifdef MCPT_REF_MAIN
// ref_render <scene.mcsd> <out.f32>: writes the raw float32 frame.
int main(int argc, char **argv)
{
    if (argc != 3)
    {
        std::fprintf(stderr, "usage: %s scene.mcsd out.f32\n", argv[0]);
        return 2;
    }
    const mcsd::Scene scene = mcsd::Load(argv[1]);
    std::vector<float> frame(static_cast<size_t>(scene.camera.width) *
                             scene.camera.height * 3);
    double seconds = 0;
    if (mcpt_ref_render(argv[1], frame.data(), &seconds) != 0)
    {
        std::fprintf(stderr, "error: %s\n", mcpt_ref_last_error());
        return 1;
    }
    FILE *f = std::fopen(argv[2], "wb");
    std::fwrite(frame.data(), sizeof(float), frame.size(), f);
    std::fclose(f);
    std::printf("{\"render_seconds\": %.6f}\n", seconds);
    return 0;
}
#endif
