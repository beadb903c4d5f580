// Neutral in-memory form of an MCSD file (see mcsd_format.h) plus a
// header-only reader and writer.  Plain structs, std::vector storage, no
// behaviour: each consumer (product host commit, oracle restatement,
// reference driver) converts from these into its own types.
#ifndef MCSD_SCENE_HPP
#define MCSD_SCENE_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "mcsd_format.h"

namespace mcsd
{

struct Camera
{
    uint32_t spp = 64;
    int32_t width = 1024, height = 1024;
    float fov_x = 19.5f;
    float eye[3] = {0, 1, 6.8f}, look_at[3] = {0, 1, 0}, up[3] = {0, 1, 0};
};

struct Integrator
{
    uint32_t type = MCSD_INTEGRATOR_PATH;
    uint32_t hide_emitters = 0;
    float pdf_rr = 0.95f;
    uint32_t depth_rr = 5;
    uint32_t depth_max = MCSD_INVALID_ID;
};

struct Texture
{
    uint32_t type = MCSD_TEX_CONSTANT;
    float color[3] = {0.5f, 0.5f, 0.5f};
    float color0[3] = {0.4f, 0.4f, 0.4f}, color1[3] = {0.2f, 0.2f, 0.2f};
    float to_uv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int32_t width = 0, height = 0, channel = 0;
    std::vector<float> data;
};

struct Bsdf
{
    uint32_t type = MCSD_BSDF_DIFFUSE;
    uint32_t twosided = 0;
    uint32_t id_opacity = MCSD_INVALID_ID, id_bump_map = MCSD_INVALID_ID;
    float weight = 1.0f;
    uint32_t id_radiance = MCSD_INVALID_ID;
    uint32_t id_diffuse_reflectance = MCSD_INVALID_ID;
    uint32_t id_roughness = MCSD_INVALID_ID;
    uint32_t use_fast_approx = 1;
    uint32_t id_roughness_u = MCSD_INVALID_ID, id_roughness_v = MCSD_INVALID_ID;
    uint32_t id_specular_reflectance = MCSD_INVALID_ID;
    uint32_t id_specular_transmittance = MCSD_INVALID_ID;
    float reflectivity[3] = {0, 0, 0}, edgetint[3] = {0, 0, 0};
    float eta = 1.0f;
};

struct Medium
{
    uint32_t type = 0;
    float sigma_a[3] = {0, 0, 0}, sigma_s[3] = {0, 0, 0};
    uint32_t phase_type = MCSD_PHASE_ISOTROPIC;
    float g[3] = {0, 0, 0};
};

struct Instance
{
    uint32_t type = MCSD_INST_MESHES;
    uint32_t id_bsdf = MCSD_INVALID_ID;
    uint32_t id_medium_int = MCSD_INVALID_ID, id_medium_ext = MCSD_INVALID_ID;
    uint32_t flip_normals = 0;
    float to_world[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float sphere_radius = 1.0f, sphere_center[3] = {0, 0, 0};
    float cyl_radius = 1.0f, cyl_p0[3] = {0, 0, 0}, cyl_p1[3] = {0, 0, 0};
    std::vector<float> texcoords;   // 2 per vertex
    std::vector<float> positions;   // 3 per vertex
    std::vector<float> normals;     // 3 per vertex
    std::vector<float> tangents;    // 3 per vertex
    std::vector<float> bitangents;  // 3 per vertex
    std::vector<uint32_t> indices;  // 3 per triangle
};

struct Emitter
{
    uint32_t type = MCSD_EMIT_DIRECTIONAL;
    float position[3] = {0, 0, 0};
    float intensity[3] = {1, 1, 1};
    float cutoff_angle = 0, beam_width = 0;
    uint32_t id_texture = MCSD_INVALID_ID;
    float to_world[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float direction[3] = {0, 0, 0};
    float radiance[3] = {0, 0, 0};
    float cos_cutoff_angle = 0;
    uint32_t id_radiance = MCSD_INVALID_ID;
};

struct Scene
{
    Camera camera;
    Integrator integrator;
    std::vector<Texture> textures;
    std::vector<Bsdf> bsdfs;
    std::vector<Medium> media;
    std::vector<Instance> instances;
    std::vector<Emitter> emitters;
};

namespace detail
{

struct Reader
{
    const uint8_t *p;
    size_t size, off = 0;
    void need(size_t n)
    {
        if (n > size - off) // off <= size always; written so that a huge n cannot wrap around
            throw std::runtime_error("MCSD: truncated file");
    }
    uint32_t u32()
    {
        need(4);
        uint32_t v;
        std::memcpy(&v, p + off, 4);
        off += 4;
        return v;
    }
    int32_t i32() { return static_cast<int32_t>(u32()); }
    float f32()
    {
        need(4);
        float v;
        std::memcpy(&v, p + off, 4);
        off += 4;
        return v;
    }
    void need_words(size_t n) // n 4-byte elements, n taken from the file
    {
        if (n > (size - off) / 4)
            throw std::runtime_error("MCSD: truncated file");
    }
    void floats(float *dst, size_t n)
    {
        need_words(n);
        std::memcpy(dst, p + off, 4 * n);
        off += 4 * n;
    }
    void fvec(std::vector<float> *dst, size_t n)
    {
        need_words(n); // before the allocation: a 100-byte file must not reserve gigabytes
        dst->resize(n);
        if (n)
            floats(dst->data(), n);
    }
};

struct Writer
{
    std::vector<uint8_t> out;
    void raw(const void *src, size_t n)
    {
        const uint8_t *s = static_cast<const uint8_t *>(src);
        out.insert(out.end(), s, s + n);
    }
    void u32(uint32_t v) { raw(&v, 4); }
    void i32(int32_t v) { raw(&v, 4); }
    void f32(float v) { raw(&v, 4); }
    void floats(const float *v, size_t n) { raw(v, 4 * n); }
    void pad_to(size_t start, size_t words)
    {
        while (out.size() < start + 4 * words)
            out.push_back(0);
    }
};

} // namespace detail

inline Scene Parse(const uint8_t *bytes, size_t size)
{
    detail::Reader r{bytes, size};
    r.need(8);
    if (std::memcmp(bytes, "MCSD", 4) != 0)
        throw std::runtime_error("MCSD: bad magic");
    r.off = 4;
    if (r.u32() != MCSD_VERSION)
        throw std::runtime_error("MCSD: unsupported version");
    Scene s;
    s.camera.spp = r.u32();
    s.camera.width = r.i32();
    s.camera.height = r.i32();
    s.camera.fov_x = r.f32();
    r.floats(s.camera.eye, 3);
    r.floats(s.camera.look_at, 3);
    r.floats(s.camera.up, 3);
    s.integrator.type = r.u32();
    s.integrator.hide_emitters = r.u32();
    s.integrator.pdf_rr = r.f32();
    s.integrator.depth_rr = r.u32();
    s.integrator.depth_max = r.u32();

    const uint32_t n_tex = r.u32();
    for (uint32_t i = 0; i < n_tex; ++i)
    {
        Texture t;
        t.type = r.u32();
        switch (t.type)
        {
        case MCSD_TEX_CONSTANT:
            r.floats(t.color, 3);
            break;
        case MCSD_TEX_CHECKERBOARD:
            r.floats(t.color0, 3);
            r.floats(t.color1, 3);
            r.floats(t.to_uv, 16);
            break;
        case MCSD_TEX_BITMAP:
            t.width = r.i32();
            t.height = r.i32();
            t.channel = r.i32();
            r.floats(t.to_uv, 16);
            if (t.width < 0 || t.height < 0 || t.channel < 0)
                throw std::runtime_error("MCSD: negative bitmap size");
            r.fvec(&t.data, static_cast<size_t>(t.width) * t.height * t.channel);
            break;
        default:
            throw std::runtime_error("MCSD: unknown texture type");
        }
        s.textures.push_back(std::move(t));
    }

    const uint32_t n_bsdf = r.u32();
    for (uint32_t i = 0; i < n_bsdf; ++i)
    {
        Bsdf b;
        b.type = r.u32();
        b.twosided = r.u32();
        b.id_opacity = r.u32();
        b.id_bump_map = r.u32();
        const size_t end = r.off + 4 * MCSD_BSDF_PAYLOAD_WORDS;
        r.need(4 * MCSD_BSDF_PAYLOAD_WORDS);
        switch (b.type)
        {
        case MCSD_BSDF_AREA_LIGHT:
            b.weight = r.f32();
            b.id_radiance = r.u32();
            break;
        case MCSD_BSDF_DIFFUSE:
            b.id_diffuse_reflectance = r.u32();
            break;
        case MCSD_BSDF_ROUGH_DIFFUSE:
            b.use_fast_approx = r.u32();
            b.id_diffuse_reflectance = r.u32();
            b.id_roughness = r.u32();
            break;
        case MCSD_BSDF_CONDUCTOR:
            b.id_roughness_u = r.u32();
            b.id_roughness_v = r.u32();
            b.id_specular_reflectance = r.u32();
            r.floats(b.reflectivity, 3);
            r.floats(b.edgetint, 3);
            break;
        case MCSD_BSDF_DIELECTRIC:
        case MCSD_BSDF_THIN_DIELECTRIC:
            b.id_roughness_u = r.u32();
            b.id_roughness_v = r.u32();
            b.id_specular_reflectance = r.u32();
            b.id_specular_transmittance = r.u32();
            b.eta = r.f32();
            break;
        case MCSD_BSDF_PLASTIC:
            b.eta = r.f32();
            b.id_roughness = r.u32();
            b.id_diffuse_reflectance = r.u32();
            b.id_specular_reflectance = r.u32();
            break;
        default:
            throw std::runtime_error("MCSD: unknown BSDF type");
        }
        r.off = end;
        s.bsdfs.push_back(b);
    }

    const uint32_t n_media = r.u32();
    for (uint32_t i = 0; i < n_media; ++i)
    {
        Medium m;
        m.type = r.u32();
        r.floats(m.sigma_a, 3);
        r.floats(m.sigma_s, 3);
        m.phase_type = r.u32();
        r.floats(m.g, 3);
        s.media.push_back(m);
    }

    const uint32_t n_inst = r.u32();
    for (uint32_t i = 0; i < n_inst; ++i)
    {
        Instance in;
        in.type = r.u32();
        in.id_bsdf = r.u32();
        in.id_medium_int = r.u32();
        in.id_medium_ext = r.u32();
        in.flip_normals = r.u32();
        r.floats(in.to_world, 16);
        in.sphere_radius = r.f32();
        r.floats(in.sphere_center, 3);
        in.cyl_radius = r.f32();
        r.floats(in.cyl_p0, 3);
        r.floats(in.cyl_p1, 3);
        uint32_t n[6];
        for (int k = 0; k < 6; ++k)
            n[k] = r.u32();
        r.fvec(&in.texcoords, 2 * static_cast<size_t>(n[0]));
        r.fvec(&in.positions, 3 * static_cast<size_t>(n[1]));
        r.fvec(&in.normals, 3 * static_cast<size_t>(n[2]));
        r.fvec(&in.tangents, 3 * static_cast<size_t>(n[3]));
        r.fvec(&in.bitangents, 3 * static_cast<size_t>(n[4]));
        r.need_words(3 * static_cast<size_t>(n[5]));
        in.indices.resize(3 * static_cast<size_t>(n[5]));
        if (!in.indices.empty())
            std::memcpy(in.indices.data(), r.p + r.off, 4 * in.indices.size());
        r.off += 4 * in.indices.size();
        s.instances.push_back(std::move(in));
    }

    const uint32_t n_emit = r.u32();
    for (uint32_t i = 0; i < n_emit; ++i)
    {
        Emitter e;
        e.type = r.u32();
        const size_t end = r.off + 4 * MCSD_EMITTER_PAYLOAD_WORDS;
        r.need(4 * MCSD_EMITTER_PAYLOAD_WORDS);
        switch (e.type)
        {
        case MCSD_EMIT_POINT:
            r.floats(e.position, 3);
            r.floats(e.intensity, 3);
            break;
        case MCSD_EMIT_SPOT:
            e.cutoff_angle = r.f32();
            e.beam_width = r.f32();
            e.id_texture = r.u32();
            r.floats(e.intensity, 3);
            r.floats(e.to_world, 16);
            break;
        case MCSD_EMIT_DIRECTIONAL:
            r.floats(e.direction, 3);
            r.floats(e.radiance, 3);
            break;
        case MCSD_EMIT_SUN:
            e.cos_cutoff_angle = r.f32();
            e.id_texture = r.u32();
            r.floats(e.direction, 3);
            r.floats(e.radiance, 3);
            break;
        case MCSD_EMIT_ENVMAP:
            e.id_radiance = r.u32();
            r.floats(e.to_world, 16);
            break;
        case MCSD_EMIT_CONSTANT:
            r.floats(e.radiance, 3);
            break;
        default:
            throw std::runtime_error("MCSD: unknown emitter type");
        }
        r.off = end;
        s.emitters.push_back(e);
    }
    if (r.off != size)
        throw std::runtime_error("MCSD: trailing bytes");
    return s;
}

inline std::vector<uint8_t> ReadFileBytes(const std::string &path)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("cannot open '" + path + "'");
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
        bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return bytes;
}

inline Scene Load(const std::string &path)
{
    const std::vector<uint8_t> bytes = ReadFileBytes(path);
    return Parse(bytes.data(), bytes.size());
}

inline std::vector<uint8_t> Serialize(const Scene &s)
{
    detail::Writer w;
    w.raw("MCSD", 4);
    w.u32(MCSD_VERSION);
    w.u32(s.camera.spp);
    w.i32(s.camera.width);
    w.i32(s.camera.height);
    w.f32(s.camera.fov_x);
    w.floats(s.camera.eye, 3);
    w.floats(s.camera.look_at, 3);
    w.floats(s.camera.up, 3);
    w.u32(s.integrator.type);
    w.u32(s.integrator.hide_emitters);
    w.f32(s.integrator.pdf_rr);
    w.u32(s.integrator.depth_rr);
    w.u32(s.integrator.depth_max);

    w.u32(static_cast<uint32_t>(s.textures.size()));
    for (const Texture &t : s.textures)
    {
        w.u32(t.type);
        switch (t.type)
        {
        case MCSD_TEX_CONSTANT:
            w.floats(t.color, 3);
            break;
        case MCSD_TEX_CHECKERBOARD:
            w.floats(t.color0, 3);
            w.floats(t.color1, 3);
            w.floats(t.to_uv, 16);
            break;
        case MCSD_TEX_BITMAP:
            w.i32(t.width);
            w.i32(t.height);
            w.i32(t.channel);
            w.floats(t.to_uv, 16);
            w.floats(t.data.data(), t.data.size());
            break;
        default:
            throw std::runtime_error("MCSD: unknown texture type");
        }
    }

    w.u32(static_cast<uint32_t>(s.bsdfs.size()));
    for (const Bsdf &b : s.bsdfs)
    {
        w.u32(b.type);
        w.u32(b.twosided);
        w.u32(b.id_opacity);
        w.u32(b.id_bump_map);
        const size_t start = w.out.size();
        switch (b.type)
        {
        case MCSD_BSDF_AREA_LIGHT:
            w.f32(b.weight);
            w.u32(b.id_radiance);
            break;
        case MCSD_BSDF_DIFFUSE:
            w.u32(b.id_diffuse_reflectance);
            break;
        case MCSD_BSDF_ROUGH_DIFFUSE:
            w.u32(b.use_fast_approx);
            w.u32(b.id_diffuse_reflectance);
            w.u32(b.id_roughness);
            break;
        case MCSD_BSDF_CONDUCTOR:
            w.u32(b.id_roughness_u);
            w.u32(b.id_roughness_v);
            w.u32(b.id_specular_reflectance);
            w.floats(b.reflectivity, 3);
            w.floats(b.edgetint, 3);
            break;
        case MCSD_BSDF_DIELECTRIC:
        case MCSD_BSDF_THIN_DIELECTRIC:
            w.u32(b.id_roughness_u);
            w.u32(b.id_roughness_v);
            w.u32(b.id_specular_reflectance);
            w.u32(b.id_specular_transmittance);
            w.f32(b.eta);
            break;
        case MCSD_BSDF_PLASTIC:
            w.f32(b.eta);
            w.u32(b.id_roughness);
            w.u32(b.id_diffuse_reflectance);
            w.u32(b.id_specular_reflectance);
            break;
        default:
            throw std::runtime_error("MCSD: unknown BSDF type");
        }
        w.pad_to(start, MCSD_BSDF_PAYLOAD_WORDS);
    }

    w.u32(static_cast<uint32_t>(s.media.size()));
    for (const Medium &m : s.media)
    {
        w.u32(m.type);
        w.floats(m.sigma_a, 3);
        w.floats(m.sigma_s, 3);
        w.u32(m.phase_type);
        w.floats(m.g, 3);
    }

    w.u32(static_cast<uint32_t>(s.instances.size()));
    for (const Instance &in : s.instances)
    {
        w.u32(in.type);
        w.u32(in.id_bsdf);
        w.u32(in.id_medium_int);
        w.u32(in.id_medium_ext);
        w.u32(in.flip_normals);
        w.floats(in.to_world, 16);
        w.f32(in.sphere_radius);
        w.floats(in.sphere_center, 3);
        w.f32(in.cyl_radius);
        w.floats(in.cyl_p0, 3);
        w.floats(in.cyl_p1, 3);
        w.u32(static_cast<uint32_t>(in.texcoords.size() / 2));
        w.u32(static_cast<uint32_t>(in.positions.size() / 3));
        w.u32(static_cast<uint32_t>(in.normals.size() / 3));
        w.u32(static_cast<uint32_t>(in.tangents.size() / 3));
        w.u32(static_cast<uint32_t>(in.bitangents.size() / 3));
        w.u32(static_cast<uint32_t>(in.indices.size() / 3));
        w.floats(in.texcoords.data(), in.texcoords.size());
        w.floats(in.positions.data(), in.positions.size());
        w.floats(in.normals.data(), in.normals.size());
        w.floats(in.tangents.data(), in.tangents.size());
        w.floats(in.bitangents.data(), in.bitangents.size());
        w.raw(in.indices.data(), 4 * in.indices.size());
    }

    w.u32(static_cast<uint32_t>(s.emitters.size()));
    for (const Emitter &e : s.emitters)
    {
        w.u32(e.type);
        const size_t start = w.out.size();
        switch (e.type)
        {
        case MCSD_EMIT_POINT:
            w.floats(e.position, 3);
            w.floats(e.intensity, 3);
            break;
        case MCSD_EMIT_SPOT:
            w.f32(e.cutoff_angle);
            w.f32(e.beam_width);
            w.u32(e.id_texture);
            w.floats(e.intensity, 3);
            w.floats(e.to_world, 16);
            break;
        case MCSD_EMIT_DIRECTIONAL:
            w.floats(e.direction, 3);
            w.floats(e.radiance, 3);
            break;
        case MCSD_EMIT_SUN:
            w.f32(e.cos_cutoff_angle);
            w.u32(e.id_texture);
            w.floats(e.direction, 3);
            w.floats(e.radiance, 3);
            break;
        case MCSD_EMIT_ENVMAP:
            w.u32(e.id_radiance);
            w.floats(e.to_world, 16);
            break;
        case MCSD_EMIT_CONSTANT:
            w.floats(e.radiance, 3);
            break;
        default:
            throw std::runtime_error("MCSD: unknown emitter type");
        }
        w.pad_to(start, MCSD_EMITTER_PAYLOAD_WORDS);
    }
    return w.out;
}

inline void Save(const Scene &s, const std::string &path)
{
    const std::vector<uint8_t> bytes = Serialize(s);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f)
        throw std::runtime_error("cannot write '" + path + "'");
    const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    if (n != bytes.size())
        throw std::runtime_error("short write to '" + path + "'");
}

} // namespace mcsd

#endif // MCSD_SCENE_HPP
