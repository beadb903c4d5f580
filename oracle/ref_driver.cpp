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
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

// The per-sample hook below needs the committed Camera / Integrator objects,
// which csrt::Renderer keeps private; open them up for this test driver only.
#define private public
#include "csrt/renderer/renderer.hpp"
#undef private
#include "csrt/renderer/bsdfs/kulla_conty.hpp"
#include "csrt/renderer/bsdfs/microfacet.hpp"
#include "csrt/renderer/renderer.hpp"
#include "csrt/rtcore/accel/bvh_builder.hpp"

#include "mcsd_scene.hpp"

// the reference-side binding shipped in integration/ (header only; the library calls
// are left out here: this driver only checks its RendererConfig -> MCSD direction)
#ifndef MCPT_BACKEND_DEMO
#define MCPT_BACKEND_NO_LIBRARY
#endif
#include "../integration/mcpt_backend.hpp"

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

// MCSD file -> csrt::RendererConfig (ToConfig above) -> MCSD bytes through the binding
// a reference maintainer would use (integration/mcpt_backend.hpp, csrt::ToMcsd).
// Returns 0 and writes the bytes to out_path.
int mcpt_ref_binding_round_trip(const char *mcsd_path, const char *out_path)
{
    try
    {
        const csrt::RendererConfig cfg = ToConfig(mcsd::Load(mcsd_path));
        mcsd::Save(csrt::ToMcsd(cfg), out_path);
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// Per-sample trace of one pixel: runs the body of DrawPixel
// (renderer.cpp:62-85) for pixel (i, j) and records, for every sample s, the
// unclamped radiance returned by Integrator::Shade and the LCG state after it.
int mcpt_ref_trace_pixel(const char *mcsd_path, uint32_t i, uint32_t j,
                         float *radiance, uint32_t *seed_after)
{
    try
    {
        const mcsd::Scene scene = mcsd::Load(mcsd_path);
        const csrt::RendererConfig cfg = ToConfig(scene);
        csrt::Renderer renderer(cfg);
        csrt::Camera *camera = renderer.camera_;
        csrt::Integrator *integrator = renderer.integrator_;
        const uint32_t pixel_offset = (j * camera->width() + i) * 3;
        uint32_t seed = csrt::Tea<4>(pixel_offset, 0);
        for (uint32_t s = 0; s < camera->spp(); ++s)
        {
            const float u = s * camera->spp_inv(),
                        v = csrt::GetVanDerCorputSequence<2>(s + 1),
                        x = 2.0f * (i + u) / camera->width() - 1.0f,
                        y = 1.0f - 2.0f * (j + v) / camera->height();
            const csrt::Vec3 look_dir = csrt::Normalize(
                camera->front() + x * camera->view_dx() + y * camera->view_dy());
            const csrt::Vec3 c = integrator->Shade(camera->eye(), look_dir, &seed);
            radiance[3 * s + 0] = c.x, radiance[3 * s + 1] = c.y, radiance[3 * s + 2] = c.z;
            seed_after[s] = seed;
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// ---- unit hooks on a committed scene --------------------------------------
// A "session" keeps one csrt::Renderer alive so that many unit calls can be
// made against its committed tables.
struct RefSession
{
    csrt::Renderer *renderer = nullptr;
};

void *mcpt_ref_open(const char *mcsd_path)
{
    try
    {
        const mcsd::Scene scene = mcsd::Load(mcsd_path);
        RefSession *s = new RefSession;
        s->renderer = new csrt::Renderer(ToConfig(scene));
        return s;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return nullptr;
    }
}

void mcpt_ref_close(void *session)
{
    RefSession *s = static_cast<RefSession *>(session);
    delete s->renderer;
    delete s;
}

// BSDF unit call.  rec_in: wo[3], wi[3], normal[3], tangent[3], bitangent[3],
// uv[2], inside (17 floats).  out: valid, pdf, attenuation[3], wi[3] (8
// floats).  mode 0 = Evaluate (bsdf.cpp:213-236), 1 = Sample (bsdf.cpp:188-211).
void mcpt_ref_bsdf(void *session, uint32_t id_bsdf, int mode, const float *rec_in,
                   uint32_t *seed, float *out)
{
    csrt::Renderer *r = static_cast<RefSession *>(session)->renderer;
    csrt::BsdfSampleRec rec;
    rec.wo = V3(rec_in), rec.wi = V3(rec_in + 3), rec.normal = V3(rec_in + 6);
    rec.tangent = V3(rec_in + 9), rec.bitangent = V3(rec_in + 12);
    rec.texcoord = csrt::Vec2{rec_in[15], rec_in[16]};
    rec.inside = rec_in[17] != 0.0f;
    if (mode == 0)
        r->bsdfs_[id_bsdf].Evaluate(&rec);
    else
        r->bsdfs_[id_bsdf].Sample(seed, &rec);
    out[0] = rec.valid ? 1.0f : 0.0f, out[1] = rec.pdf;
    out[2] = rec.attenuation.x, out[3] = rec.attenuation.y, out[4] = rec.attenuation.z;
    out[5] = rec.wi.x, out[6] = rec.wi.y, out[7] = rec.wi.z;
}

// Closest-hit unit call (tlas.cpp:13-42).  out: valid, inside, id_instance,
// id_primitive (as floats), t, uv[2], position[3], normal[3], tangent[3],
// bitangent[3] (19 floats).
void mcpt_ref_intersect(void *session, const float *origin, const float *dir,
                        uint32_t *seed, float *out)
{
    csrt::Renderer *r = static_cast<RefSession *>(session)->renderer;
    csrt::Ray ray(V3(origin), V3(dir));
    csrt::Integrator *integ = r->integrator_;
    const csrt::Hit hit = integ->data_.tlas->Intersect(
        integ->data_.bsdfs, integ->data_.map_instance_bsdf, seed, &ray);
    out[0] = hit.valid, out[1] = hit.inside, out[2] = static_cast<float>(hit.id_instance == csrt::kInvalidId ? -1.0 : hit.id_instance);
    out[3] = static_cast<float>(hit.id_primitve == csrt::kInvalidId ? -1.0 : hit.id_primitve);
    out[4] = ray.t_max, out[5] = hit.texcoord.u, out[6] = hit.texcoord.v;
    const csrt::Vec3 v[4] = {hit.position, hit.normal, hit.tangent, hit.bitangent};
    for (int k = 0; k < 4; ++k)
        out[7 + 3 * k] = v[k].x, out[8 + 3 * k] = v[k].y, out[9 + 3 * k] = v[k].z;
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

#ifdef MCPT_BACKEND_DEMO
// backend_demo <scene.mcsd> <out.f32>: what the reference's RayTracer would do with the
// binding of integration/mcpt_backend.hpp — RendererConfig in, frame out, on the GPU
// through libmcpt_hip.so.  Built by `make ref` next to the library; the GPU tests run it.
int main(int argc, char **argv)
{
    if (argc != 3)
    {
        std::fprintf(stderr, "usage: %s scene.mcsd out.f32\n", argv[0]);
        return 2;
    }
    try
    {
        const csrt::RendererConfig config = ToConfig(mcsd::Load(argv[1]));
        std::vector<float> frame(static_cast<size_t>(config.camera.width) * config.camera.height * 3);
        csrt::HipBackend backend(config, 0);
        backend.Draw(frame.data());
        FILE *f = std::fopen(argv[2], "wb");
        std::fwrite(frame.data(), sizeof(float), frame.size(), f);
        std::fclose(f);
        return 0;
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
#endif
