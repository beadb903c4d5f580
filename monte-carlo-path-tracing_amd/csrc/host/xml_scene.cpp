// Mitsuba-style XML scene -> renderer configuration (mcsd::Scene).
//
// Replaces the reference front end csrt::LoadConfig (src/parser/parser.cpp:
// 94-1617) for the element types its example scenes use.  The translation
// rules — element order, defaults, id assignment, the quirks — follow the
// reference so that the same XML yields the same configuration:
//   * pass order: defaults, sensor, integrator, textures, bsdfs, media, shapes,
//     emitters (parser.cpp:126-176);
//   * every BSDF parameter becomes its own texture record, created in the order
//     the reference reads them (parser.cpp:827-1006);
//   * "smooth" dielectric / conductor / plastic are the rough models with
//     alpha = 0.001 (parser.cpp:896-898, 935-937, 970-971);
//   * a shape with an <emitter> child gets an area-light pseudo-BSDF
//     (parser.cpp:1068-1100); otherwise an inline <bsdf>, otherwise the id of the
//     FIRST <ref> child is looked up among the BSDFs (parser.cpp:1108-1117) —
//     when that first ref names a medium the shape has no BSDF;
//   * only "fovAxis" and "focalLength" are recognised as <string> children of the
//     sensor (parser.cpp:262-279), so a snake-case "fov_axis" is ignored;
//   * a 9-number <matrix> is read with the reference's off-by-one
//     (parser.cpp:1545-1549);
//   * integer values go through stoi, so max_depth = -1 becomes 0xFFFFFFFF.
// Not supported (reported as errors): sun / sky emitters (Hosek-Wilkie model),
// JPEG and other stb-only bitmap formats, glTF meshes.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <vector>
#include <cstdlib>
#include <stdexcept>

#include "../vecmath.h"
#include "matrix.hpp"
#include "asset_io.hpp"
#include "standin_mesh.hpp"
#include "frontend.hpp"
#include "xml_dom.hpp"

namespace mcpt
{
namespace
{

using xml::Node;

// ---- transforms of the scene description (reference src/tensor/mat4.cpp) -----
constexpr uint32_t kNone = MCSD_INVALID_ID;

Mat4f ScaleMatrix(V3 s) // mat4.cpp:221-227
{
    Mat4f r = Identity();
    r.m[0] = s.x, r.m[5] = s.y, r.m[10] = s.z;
    return r;
}
Mat4f RotationMatrix(float angle, V3 axis) // mat4.cpp:229-249
{
    const float c = cosf(angle), s = sinf(angle);
    axis = normalize(axis);
    const V3 t = (1.0f - c) * axis;
    Mat4f r = Identity();
    r.m[0] = c + t.x * axis.x, r.m[1] = t.y * axis.x - s * axis.z, r.m[2] = t.z * axis.x + s * axis.y;
    r.m[4] = t.x * axis.y + s * axis.z, r.m[5] = c + t.y * axis.y, r.m[6] = t.z * axis.y - s * axis.x;
    r.m[8] = t.x * axis.z - s * axis.y, r.m[9] = t.y * axis.z + s * axis.x, r.m[10] = c + t.z * axis.z;
    return r;
}
Mat4f LookAtLeftHanded(V3 eye, V3 target, V3 up) // mat4.cpp:251-261
{
    const V3 front = normalize(target - eye), right = normalize(cross(up, front));
    up = normalize(cross(front, right));
    Mat4f r = Identity();
    r.m[0] = right.x, r.m[1] = right.y, r.m[2] = right.z, r.m[3] = -dot(right, eye);
    r.m[4] = up.x, r.m[5] = up.y, r.m[6] = up.z, r.m[7] = -dot(up, eye);
    r.m[8] = front.x, r.m[9] = front.y, r.m[10] = front.z, r.m[11] = -dot(front, eye);
    return r;
}
float Radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// ---- named material data (reference src/parser/ior_lut.cpp, medium_lut.cpp) -----
#include "material_tables.inc"

// "name | n n n | n n n ..." lines -> (name, numbers) records.  Numbers are read as
// float literals (strtof), like the constants of the reference's tables.
struct MaterialRecord
{
    std::string name;
    std::vector<float> values;
};
std::vector<MaterialRecord> ParseMaterialTable(const char *text)
{
    std::vector<MaterialRecord> records;
    std::istringstream lines(text);
    std::string line;
    while (std::getline(lines, line))
    {
        const size_t bar = line.find('|');
        if (bar == std::string::npos)
            continue;
        MaterialRecord r;
        r.name = line.substr(0, bar);
        while (!r.name.empty() && r.name.back() == ' ')
            r.name.pop_back();
        const char *p = line.c_str() + bar + 1;
        for (;;)
        {
            while (*p == ' ' || *p == '|')
                ++p;
            if (!*p)
                break;
            char *end = nullptr;
            r.values.push_back(std::strtof(p, &end));
            if (end == p)
                break;
            p = end;
        }
        records.push_back(std::move(r));
    }
    return records;
}
const MaterialRecord *FindMaterial(const std::vector<MaterialRecord> &table, const std::string &name)
{
    for (const MaterialRecord &r : table)
        if (r.name == name)
            return &r;
    return nullptr;
}

bool LookupDielectricIor(const std::string &name, float *ior) // ior_lut.cpp:252-264
{
    static const std::vector<MaterialRecord> table = ParseMaterialTable(kDielectricIorText);
    const MaterialRecord *r = FindMaterial(table, name);
    if (r)
        *ior = r->values.at(0);
    return r != nullptr;
}
bool LookupConductorIor(const std::string &name, V3 *eta, V3 *k) // ior_lut.cpp:266-278
{
    static const std::vector<MaterialRecord> table = ParseMaterialTable(kConductorIorText);
    const MaterialRecord *r = FindMaterial(table, name);
    if (r)
        *eta = V3{r->values.at(0), r->values.at(1), r->values.at(2)}, *k = V3{r->values.at(3), r->values.at(4), r->values.at(5)};
    return r != nullptr;
}
// medium_lut.cpp:191-219: measured media (with g) first, then the isotropic fits
bool LookupMedium(const std::string &name, V3 *sigma_a, V3 *sigma_s, V3 *g, bool *has_g)
{
    static const std::vector<MaterialRecord> measured = ParseMaterialTable(kMeasuredMediaText),
                                             isotropic = ParseMaterialTable(kIsotropicMediaText);
    const MaterialRecord *r = FindMaterial(measured, name);
    *has_g = r != nullptr;
    if (!r)
        r = FindMaterial(isotropic, name);
    if (!r)
        return false;
    *sigma_s = V3{r->values.at(0), r->values.at(1), r->values.at(2)};
    *sigma_a = V3{r->values.at(3), r->values.at(4), r->values.at(5)};
    if (*has_g)
        *g = V3{r->values.at(6), r->values.at(7), r->values.at(8)};
    return true;
}

class SceneBuilder
{
public:
    SceneBuilder(const std::string &path, StandinTable standins) : directory_(DirectoryOf(path)), standins_(std::move(standins)) {}

    mcsd::Scene Run(const Node &scene)
    {
        for (const auto &c : scene.children) // parser.cpp:126-133
            if (c->name == "default")
                defaults_["$" + c->Str("name")] = c->Str("value");
        const Node *sensor = scene.Child("sensor");
        if (!sensor)
            throw std::runtime_error("[error] only support 'perspective' sensor.");
        ReadCamera(*sensor);
        static const Node empty;
        const Node *integrator = scene.Child("integrator");
        ReadIntegrator(integrator ? *integrator : empty);
        for (const auto &c : scene.children)
            if (c->name == "texture")
                ReadTexture(c.get(), 1.0f, 1.0f);
        for (const auto &c : scene.children)
            if (c->name == "bsdf")
                ReadBsdf(*c, "", kNone, kNone, false);
        for (const auto &c : scene.children)
            if (c->name == "medium")
                ReadMedium(*c);
        for (const auto &c : scene.children)
            if (c->name == "shape")
                ReadShape(*c);
        for (const auto &c : scene.children)
            if (c->name == "emitter")
                ReadEmitter(*c);
        return std::move(out_);
    }

private:
    static std::string DirectoryOf(const std::string &path)
    {
        const size_t slash = path.find_last_of("/\\");
        return slash == std::string::npos ? "" : path.substr(0, slash + 1);
    }

    // ---- basic readers (parser.cpp:1430-1617) --------------------------------
    static const Node *ChildByName(const Node &parent, std::initializer_list<const char *> names)
    {
        for (const char *name : names)
            for (const auto &c : parent.children)
                if (c->Str("name") == name)
                    return c.get();
        return nullptr;
    }
    static bool ReadBool(const Node &parent, std::initializer_list<const char *> names, bool fallback)
    {
        const Node *n = ChildByName(parent, names);
        return n ? n->Bool("value", fallback) : fallback;
    }
    static float ReadFloat(const Node &parent, std::initializer_list<const char *> names, float fallback)
    {
        const Node *n = ChildByName(parent, names);
        return n ? n->Float("value", fallback) : fallback;
    }
    static int ReadInt(const Node &parent, std::initializer_list<const char *> names, int fallback)
    {
        const Node *n = ChildByName(parent, names);
        return n ? n->Int("value", fallback) : fallback;
    }
    // parser.cpp:1488-1530
    static V3 ReadVec3Node(const Node &n, V3 fallback, std::string value_name)
    {
        if (!n.Has("value") && value_name.empty())
            return V3{n.Float("x", fallback.x), n.Float("y", fallback.y), n.Float("z", fallback.z)};
        if (value_name.empty())
            value_name = "value";
        const std::string text = n.Str(value_name);
        int spaces = 0, commas = 0;
        for (char c : text)
            spaces += c == ' ', commas += c == ',';
        if (spaces == 0)
            return splat(n.Float("value", fallback.x)); // reads "value" whatever value_name is
        if (spaces == 2)
        {
            V3 r{0, 0, 0};
            if (commas == 0)
                std::sscanf(text.c_str(), "%f %f %f", &r.x, &r.y, &r.z);
            else
                std::sscanf(text.c_str(), "%f, %f, %f", &r.x, &r.y, &r.z);
            return r;
        }
        return fallback;
    }
    static V3 ReadVec3(const Node &parent, std::initializer_list<const char *> names, V3 fallback)
    {
        const Node *n = ChildByName(parent, names);
        return n ? ReadVec3Node(*n, fallback, "") : fallback;
    }
    static Mat4f ReadMatrix(const Node &n) // parser.cpp:1532-1561
    {
        Mat4f r = Identity();
        if (!n.Has("value"))
            return r;
        const std::string text = n.Str("value");
        int spaces = 0;
        for (char c : text)
            spaces += c == ' ';
        if (spaces == 8)
        {
            // nine numbers land in [0][0] (twice), [0][1], [1][0..2], [2][0..2]
            float first;
            std::sscanf(text.c_str(), "%f %f %f %f %f %f %f %f %f", &first, &r.m[0], &r.m[1], &r.m[4], &r.m[5], &r.m[6],
                        &r.m[8], &r.m[9], &r.m[10]);
        }
        else
        {
            float *m = r.m;
            std::sscanf(text.c_str(), "%f %f %f %f %f %f %f %f %f %f %f %f %f %f %f %f", m, m + 1, m + 2, m + 3, m + 4,
                        m + 5, m + 6, m + 7, m + 8, m + 9, m + 10, m + 11, m + 12, m + 13, m + 14, m + 15);
        }
        return r;
    }
    static Mat4f ReadTransform(const Node *t) // parser.cpp:1563-1617
    {
        Mat4f result = Identity();
        if (!t)
            return result;
        for (const auto &c : t->children)
        {
            if (c->name == "translate")
                result = Multiply(TranslationMatrix(ReadVec3Node(*c, V3{0, 0, 0}, "")), result);
            else if (c->name == "rotate")
                result = Multiply(RotationMatrix(Radians(c->Float("angle", 0)), ReadVec3Node(*c, V3{0, 0, 0}, "")), result);
            else if (c->name == "scale")
                result = Multiply(ScaleMatrix(ReadVec3Node(*c, V3{1, 1, 1}, "")), result);
            else if (c->name == "matrix")
                result = Multiply(ReadMatrix(*c), result);
            else if (c->name == "lookat")
            {
                const V3 origin = ReadVec3Node(*c, V3{0, 0, 0}, "origin"), target = ReadVec3Node(*c, V3{1, 0, 0}, "target"),
                         up = ReadVec3Node(*c, V3{0, 1, 0}, "up");
                result = Multiply(Inverted(LookAtLeftHanded(origin, target, up)), result);
            }
            else
                std::fprintf(stderr, "[warning] unsupport transform type '%s', ignore it.\n", c->name.c_str());
        }
        return result;
    }

    std::string Substitute(std::string value, const char *what) const
    {
        if (!value.empty() && value[0] == '$')
        {
            const auto it = defaults_.find(value);
            if (it == defaults_.end())
                throw std::runtime_error(std::string("cannot find default value for ") + what + ".");
            value = it->second;
        }
        return value;
    }

    // ---- sensor (parser.cpp:183-357) ------------------------------------------
    void ReadCamera(const Node &sensor)
    {
        if (sensor.Str("type") != "perspective")
            throw std::runtime_error("[error] only support 'perspective' sensor.");
        int width = 768, height = 576;
        if (const Node *film = sensor.Child("film"))
            for (const auto &c : film->children)
                if (c->name == "integer")
                {
                    const std::string name = c->Str("name"), value = c->Str("value");
                    if (value.empty())
                        continue;
                    if (name == "width")
                        width = std::stoi(Substitute(value, "width"));
                    else if (name == "height")
                        height = std::stoi(Substitute(value, "height"));
                }
        out_.camera.width = width, out_.camera.height = height;

        float focal_length = 50.0f;
        std::string fov_axis = "x";
        for (const auto &c : sensor.children)
            if (c->name == "string")
            {
                if (c->Str("name") == "focalLength")
                {
                    std::string s = c->Str("value");
                    s = s.substr(0, s.size() >= 2 ? s.size() - 2 : 0);
                    focal_length = std::stof(s);
                }
                else if (c->Str("name") == "fovAxis")
                    fov_axis = c->Str("value");
            }
        float fov_x = ReadFloat(sensor, {"fov"}, -1.0f);
        if (fov_axis == "x")
        {
            if (fov_x <= 0.0f)
                fov_x = 2.0f * atanf(36.0f * 0.5f / focal_length) * 180.0f * k1DivPi;
        }
        else if (fov_axis == "y")
        {
            if (fov_x <= 0.0f)
                fov_x = 2.0f * atanf(24.0f * 0.5f / focal_length) * 180.0f * k1DivPi;
            fov_x = fov_x * width / height;
        }
        else if (fov_axis == "smaller")
        {
            if (width > height)
            {
                if (fov_x <= 0.0f)
                    fov_x = 2.0f * atanf(24.0f * 0.5f / focal_length) * 180.0f * k1DivPi;
                fov_x = fov_x * width / height;
            }
        }
        else
            throw std::runtime_error("[error] unsupport fov axis type '" + fov_axis + "'");
        out_.camera.fov_x = fov_x;

        int spp = 4;
        if (const Node *sampler = sensor.Child("sampler"))
            for (const auto &c : sampler->children)
                if (c->name == "integer" && (c->Str("name") == "sampleCount" || c->Str("name") == "sample_count"))
                {
                    const std::string value = c->Str("value");
                    if (!value.empty())
                        spp = std::stoi(Substitute(value, "sample count"));
                }
        out_.camera.spp = static_cast<uint32_t>(spp);

        V3 eye{0, 0, 0}, look_at{0, 0, 1}, up{0, 1, 0};
        if (const Node *t = sensor.Child("transform"))
        {
            const Mat4f to_world = (ReadTransform(t));
            eye = transform_point(to_world, eye);
            look_at = transform_point(to_world, look_at);
            up = transform_dir(to_world, up);
        }
        out_.camera.eye[0] = eye.x, out_.camera.eye[1] = eye.y, out_.camera.eye[2] = eye.z;
        out_.camera.look_at[0] = look_at.x, out_.camera.look_at[1] = look_at.y, out_.camera.look_at[2] = look_at.z;
        out_.camera.up[0] = up.x, out_.camera.up[1] = up.y, out_.camera.up[2] = up.z;
    }

    // ---- integrator (parser.cpp:359-441) ----------------------------------------
    void ReadIntegrator(const Node &n)
    {
        uint32_t max_depth = kNone, rr_depth = 5;
        for (const auto &c : n.children)
            if (c->name == "integer")
            {
                const std::string name = c->Str("name"), value = c->Str("value");
                if (value.empty())
                    continue;
                if (name == "maxDepth" || name == "max_depth")
                    max_depth = static_cast<uint32_t>(std::stoi(Substitute(value, "max depth")));
                else if (name == "rrDepth" || name == "rr_depth")
                    rr_depth = static_cast<uint32_t>(std::stoi(Substitute(value, "rr depth")));
            }
        out_.integrator.depth_max = max_depth;
        out_.integrator.depth_rr = rr_depth;
        out_.integrator.hide_emitters = ReadBool(n, {"hide_emitters", "hideEmitters"}, false);
        out_.integrator.pdf_rr = ReadFloat(n, {"rr_pdf", "rrPdf"}, 0.95f);
        std::string type = n.Str("type", "path");
        if (!type.empty() && type[0] == '$' && defaults_.count(type))
            type = defaults_.at(type);
        if (type == "volpath")
            out_.integrator.type = MCSD_INTEGRATOR_VOLPATH;
        else
        {
            if (type != "path")
                std::fprintf(stderr, "unsupport integrator type '%s', use 'path' instead.\n", type.c_str());
            out_.integrator.type = MCSD_INTEGRATOR_PATH;
        }
    }

    // ---- textures (parser.cpp:443-671) -------------------------------------------
    uint32_t AddConstant(const std::string &id_hint, V3 color)
    {
        const uint32_t index = static_cast<uint32_t>(out_.textures.size());
        const std::string id = id_hint.empty() ? "texture_" + std::to_string(index) : id_hint;
        texture_ids_[id] = index;
        mcsd::Texture t;
        t.type = MCSD_TEX_CONSTANT;
        t.color[0] = color.x, t.color[1] = color.y, t.color[2] = color.z;
        out_.textures.push_back(t);
        return index;
    }
    uint32_t AddBitmap(const std::string &path, const std::string &id, float gamma, float scale, const int *width_max)
    {
        ImageData img = LoadFloatImage(path, gamma);
        if (width_max && img.width > *width_max && *width_max > 0) // image_io.cpp:160-169
            img = ResizeLikeReference(img, *width_max, *width_max * img.height / img.width);
        for (float &v : img.data)
            v *= scale;
        const uint32_t index = static_cast<uint32_t>(out_.textures.size());
        texture_ids_[id] = index;
        mcsd::Texture t;
        t.type = MCSD_TEX_BITMAP;
        t.width = img.width, t.height = img.height, t.channel = img.channel;
        t.data = std::move(img.data);
        out_.textures.push_back(std::move(t));
        return index;
    }
    uint32_t ReadTexture(const Node *n, float scale, float fallback)
    {
        if (!n)
            return AddConstant("", splat(scale * fallback));
        std::string id = n->Str("id");
        if (n->name == "scale")
            return ReadTexture(n->Child("texture"), scale * ReadFloat(*n, {"scale"}, 1.0f), fallback);
        if (n->name == "ref")
        {
            const auto it = texture_ids_.find(id);
            if (it == texture_ids_.end())
                throw std::runtime_error("[error] cannot find texture with id '" + id + "'.");
            return it->second;
        }
        if (n->name == "rgb")
            return AddConstant(id, scale * ReadVec3Node(*n, splat(fallback), ""));
        if (n->name == "float")
            return AddConstant(id, splat(scale * n->Float("value", fallback)));
        if (n->name != "texture")
            throw std::runtime_error("[error] unsupport texture type '" + n->name + "'.");
        const std::string type = n->Str("type");
        if (type == "checkerboard")
        {
            auto colour = [&](const char *name, float dflt)
            {
                const Node *c = ChildByName(*n, {name});
                if (!c)
                    return splat(dflt);
                if (c->name != "rgb" && c->name != "float")
                    throw std::runtime_error("not support texture inside 'checkerboard'");
                return ReadVec3(*n, {name}, splat(dflt));
            };
            const V3 color0 = colour("color0", 0.4f), color1 = colour("color1", 0.2f);
            Mat4f to_uv = ReadTransform(n->Child("transform"));
            const float uo = ReadFloat(*n, {"uoffset"}, 0.0f), vo = ReadFloat(*n, {"voffset"}, 0.0f),
                        us = ReadFloat(*n, {"uscale"}, 1.0f), vs = ReadFloat(*n, {"vscale"}, 1.0f);
            to_uv = Multiply(TranslationMatrix(V3{uo, vo, 0.0f}), to_uv);
            to_uv = Multiply(ScaleMatrix(V3{us, vs, 1.0f}), to_uv);
            const uint32_t index = static_cast<uint32_t>(out_.textures.size());
            if (id.empty())
                id = "texture_" + std::to_string(index);
            texture_ids_[id] = index;
            mcsd::Texture t;
            t.type = MCSD_TEX_CHECKERBOARD;
            const V3 c0 = scale * color0, c1 = scale * color1;
            t.color0[0] = c0.x, t.color0[1] = c0.y, t.color0[2] = c0.z;
            t.color1[0] = c1.x, t.color1[1] = c1.y, t.color1[2] = c1.z;
            std::memcpy(t.to_uv, to_uv.m, sizeof(t.to_uv));
            out_.textures.push_back(t);
            return index;
        }
        if (type == "bitmap")
        {
            const Node *file = ChildByName(*n, {"filename"});
            if (!file)
                throw std::runtime_error("[error] cannot find filename for bitmap texture.");
            const float gamma = ReadFloat(*n, {"gamma"}, 0.0f);
            if (id.empty())
                id = "texture_" + std::to_string(out_.textures.size());
            return AddBitmap(directory_ + file->Str("value"), id, gamma, scale, nullptr);
        }
        throw std::runtime_error("[error] unsupport texture type '" + type + "'.");
    }
    uint32_t ReadTextureParam(const Node &parent, std::initializer_list<const char *> names, float fallback)
    {
        const Node *n = names.size() ? ChildByName(parent, names) : nullptr;
        if (n)
            return ReadTexture(n, 1.0f, fallback);
        return AddConstant("", splat(fallback));
    }

    // ---- media (parser.cpp:673-788) -------------------------------------------------
    uint32_t ReadMedium(const Node &n)
    {
        std::string id = n.Str("id");
        const auto it = medium_ids_.find(id);
        if (it != medium_ids_.end())
            return it->second;
        const uint32_t index = static_cast<uint32_t>(out_.media.size());
        if (id.empty())
            id = "medium_" + std::to_string(index);
        medium_ids_[id] = index;
        if (n.Str("type") != "homogeneous")
            throw std::runtime_error("unsupported  media'" + n.Str("type") + "'.");
        const float scale = ReadFloat(n, {"scale"}, 1.0f);
        V3 sigma_s{0, 0, 0}, sigma_a{0, 0, 0};
        const Node *albedo_node = ChildByName(n, {"albedo"});
        if (albedo_node)
        {
            const Node *st = ChildByName(n, {"sigma_t", "sigmaT"});
            if (!st)
                throw std::runtime_error("'sigma_t' and 'albedo' must be provided at the same time.");
            const V3 albedo = ReadVec3Node(*albedo_node, splat(0.75f), ""), sigma_t = ReadVec3Node(*st, splat(1.0f), "");
            sigma_s = albedo * sigma_t;
            sigma_a = sigma_t - sigma_s;
        }
        const Node *sa = ChildByName(n, {"sigmaA"});
        if (sa)
        {
            const Node *ss = ChildByName(n, {"sigmaS"});
            if (!ss)
                throw std::runtime_error("'sigma_a' and 'sigma_s' must be provided at the same time.");
            sigma_a = ReadVec3Node(*sa, splat(1.0f), "");
            sigma_s = ReadVec3Node(*ss, splat(1.0f), "");
        }
        mcsd::Medium m;
        auto store = [&](V3 a, V3 sc)
        {
            a = a * scale, sc = sc * scale;
            m.sigma_a[0] = a.x, m.sigma_a[1] = a.y, m.sigma_a[2] = a.z;
            m.sigma_s[0] = sc.x, m.sigma_s[1] = sc.y, m.sigma_s[2] = sc.z;
        };
        m.phase_type = MCSD_PHASE_ISOTROPIC;
        if (!albedo_node && !sa)
        {
            // named medium (parser.cpp:731-757); the default name "skin1" is in neither table
            const Node *name_node = n.Child("string");
            const std::string name = name_node ? name_node->Str("value", "skin1") : "skin1";
            V3 table_a, table_s, table_g;
            bool has_g = false;
            const bool found = LookupMedium(name, &table_a, &table_s, &table_g, &has_g);
            if (found)
            {
                store(table_a, table_s);
                if (has_g)
                {
                    m.phase_type = MCSD_PHASE_HG;
                    m.g[0] = table_g.x, m.g[1] = table_g.y, m.g[2] = table_g.z;
                }
            }
            if (!found)
                throw std::runtime_error("unsupport medium type '" + name + "'.");
        }
        else
        {
            store(sigma_a, sigma_s);
            if (const Node *phase = n.Child("phase"))
            {
                const std::string type = phase->Str("type");
                if (type == "hg")
                {
                    m.phase_type = MCSD_PHASE_HG;
                    const float g = ReadFloat(*phase, {"g"}, 0);
                    m.g[0] = m.g[1] = m.g[2] = g;
                }
                else if (type != "isotropic")
                    std::fprintf(stderr, "[warning] unsupport phase function '%s',  use 'isotropic' instead.\n",
                                 type.c_str());
            }
        }
        out_.media.push_back(m);
        return index;
    }

    // ---- BSDFs (parser.cpp:790-1066) ------------------------------------------------
    float ReadDielectricIor(const Node &parent, std::initializer_list<const char *> names, float fallback)
    {
        const Node *n = ChildByName(parent, names);
        if (n && n->name == "string")
        {
            float ior = 0;
            if (!LookupDielectricIor(n->Str("value"), &ior))
                throw std::runtime_error("unsupported  material'" + n->Str("value") + "'.");
            return ior;
        }
        return n ? n->Float("value", fallback) : fallback;
    }
    void ReadConductorIor(const Node &parent, V3 *eta, V3 *k)
    {
        if (const Node *m = ChildByName(parent, {"material"}))
        {
            if (!LookupConductorIor(m->Str("value"), eta, k))
                throw std::runtime_error("unsupported  material'" + m->Str("value") + "'.");
        }
        else if (const Node *e = ChildByName(parent, {"eta"}))
        {
            *eta = ReadVec3Node(*e, V3{1, 1, 1}, "");
            const Node *kn = ChildByName(parent, {"k"});
            if (!kn)
                throw std::runtime_error("cannot find 'k for Conductor bsdf'" + parent.Str("id") + "'.");
            *k = ReadVec3Node(*kn, V3{1, 1, 1}, "");
        }
        else
            LookupConductorIor("Cu", eta, k);
    }
    uint32_t ReadBsdf(const Node &n, std::string id, uint32_t id_opacity, uint32_t id_bump, bool twosided)
    {
        if (id.empty())
            id = n.Str("id");
        const std::string type = n.Str("type");
        auto inner = [&]() -> const Node &
        {
            // a wrapper without a nested <bsdf> reads an empty node, as the reference does
            static const Node empty;
            const Node *c = n.Child("bsdf");
            return c ? *c : empty;
        };
        if (type == "bumpmap")
            return ReadBsdf(inner(), id, id_opacity, ReadTexture(n.Child("texture"), 1.0f, 1.0f), twosided);
        if (type == "mask")
            return ReadBsdf(inner(), id, ReadTextureParam(n, {"opacity"}, 1.0f), id_bump, twosided);
        if (type == "twosided")
            return ReadBsdf(inner(), id, id_opacity, id_bump, true);
        for (const char *unsupported : {"coating", "roughcoating", "phong", "ward", "mixturebsdf", "blendbsdf",
                                        "difftrans", "hk", "irawan", "null"})
            if (type == unsupported)
                throw std::runtime_error("[error] not support bsdf type '" + type + "'.");

        const uint32_t index = static_cast<uint32_t>(out_.bsdfs.size());
        if (id.empty())
            id = "bsdf_" + std::to_string(index);
        mcsd::Bsdf b;
        b.twosided = twosided, b.id_opacity = id_opacity, b.id_bump_map = id_bump;
        auto roughness_pair = [&](bool rough, uint32_t *ru, uint32_t *rv)
        {
            if (rough)
            {
                if (ChildByName(n, {"alpha"}))
                    *ru = *rv = ReadTextureParam(n, {"alpha"}, 0.1f);
                else
                {
                    *ru = ReadTextureParam(n, {"alpha_u", "alphaU"}, 0.1f);
                    *rv = ReadTextureParam(n, {"alpha_v", "alphaV"}, 0.1f);
                }
            }
            else
                *ru = *rv = ReadTextureParam(n, {}, 0.001f);
        };
        if (type == "diffuse")
        {
            b.type = MCSD_BSDF_DIFFUSE;
            b.id_diffuse_reflectance = ReadTextureParam(n, {"reflectance"}, 0.5f);
        }
        else if (type == "roughdiffuse")
        {
            b.type = MCSD_BSDF_ROUGH_DIFFUSE;
            b.use_fast_approx = ReadBool(n, {"useFastApprox"}, false);
            b.id_diffuse_reflectance = ReadTextureParam(n, {"reflectance"}, 0.5f);
            b.id_roughness = ReadTextureParam(n, {"alpha"}, 0.2f);
        }
        else if (type == "dielectric" || type == "roughdielectric" || type == "thindielectric")
        {
            b.twosided = 1;
            const float int_ior = ReadDielectricIor(n, {"int_ior", "intIOR"}, 1.5046f),
                        ext_ior = ReadDielectricIor(n, {"ext_ior", "extIOR"}, 1.000277f);
            roughness_pair(type == "roughdielectric", &b.id_roughness_u, &b.id_roughness_v);
            b.id_specular_reflectance = ReadTextureParam(n, {"specularReflectance", "specular_reflectance"}, 1.0f);
            b.id_specular_transmittance = ReadTextureParam(n, {"specularTransmittance", "specular_transmittance"}, 1.0f);
            b.type = type == "thindielectric" ? MCSD_BSDF_THIN_DIELECTRIC : MCSD_BSDF_DIELECTRIC;
            b.eta = int_ior / ext_ior;
        }
        else if (type == "conductor" || type == "roughconductor")
        {
            roughness_pair(type == "roughconductor", &b.id_roughness_u, &b.id_roughness_v);
            b.id_specular_reflectance = ReadTextureParam(n, {"specularReflectance", "specular_reflectance"}, 1.0f);
            V3 eta, k;
            ReadConductorIor(n, &eta, &k);
            // parser.cpp:944-949
            const V3 r = (sqr(eta - 1.0f) + sqr(k)) / (sqr(eta + 1.0f) + sqr(k));
            const V3 t1 = 1.0f + vsqrt(r), t2 = 1.0f - vsqrt(r), t3 = (1.0f - r) / (1.0f + r);
            const V3 edgetint = (t1 - eta * t2) / (t1 - t3 * t2);
            b.type = MCSD_BSDF_CONDUCTOR;
            b.reflectivity[0] = r.x, b.reflectivity[1] = r.y, b.reflectivity[2] = r.z;
            b.edgetint[0] = edgetint.x, b.edgetint[1] = edgetint.y, b.edgetint[2] = edgetint.z;
        }
        else if (type == "plastic" || type == "roughplastic")
        {
            const float int_ior = ReadDielectricIor(n, {"int_ior", "intIOR"}, 1.5046f),
                        ext_ior = ReadDielectricIor(n, {"ext_ior", "extIOR"}, 1.000277f);
            b.id_roughness = type == "roughplastic" ? ReadTextureParam(n, {"alpha"}, 0.1f) : ReadTextureParam(n, {}, 0.001f);
            b.id_diffuse_reflectance = ReadTextureParam(n, {"diffuseReflectance", "diffuse_reflectance"}, 1.0f);
            b.id_specular_reflectance = ReadTextureParam(n, {"specularReflectance", "specular_reflectance"}, 1.0f);
            b.type = MCSD_BSDF_PLASTIC;
            b.eta = int_ior / ext_ior;
        }
        else
        {
            std::fprintf(stderr, "[warning] unsupport bsdf type '%s', use default 'diffuse' instead.\n", type.c_str());
            b.type = MCSD_BSDF_DIFFUSE;
            b.id_diffuse_reflectance = ReadTextureParam(n, {}, 0.5f);
        }
        out_.bsdfs.push_back(b);
        bsdf_ids_[id] = index;
        return index;
    }

    // ---- shapes (parser.cpp:1058-1222) ------------------------------------------------
    void ReadShape(const Node &n)
    {
        std::string id = n.Str("id");
        if (id.empty())
            id = "shape_" + std::to_string(out_.instances.size());
        uint32_t id_bsdf = kNone;
        if (const Node *emitter = n.Child("emitter"))
        {
            if (!emitter->Child("rgb"))
                throw std::runtime_error("cannot find radiance for area light '" + id + "'.");
            const V3 radiance = ReadVec3(*emitter, {"radiance"}, V3{1, 1, 1});
            mcsd::Bsdf light;
            light.type = MCSD_BSDF_AREA_LIGHT;
            light.twosided = 0;
            light.weight = 1.0f;
            light.id_radiance = AddConstant("", radiance);
            id_bsdf = static_cast<uint32_t>(out_.bsdfs.size());
            out_.bsdfs.push_back(light);
            bsdf_ids_[id] = id_bsdf;
        }
        else if (const Node *bsdf = n.Child("bsdf"))
            id_bsdf = ReadBsdf(*bsdf, "", kNone, kNone, false);
        else if (const Node *first_ref = n.Child("ref"))
        {
            // only the FIRST <ref> child's id is ever looked up (reference quirk)
            const auto it = bsdf_ids_.find(first_ref->Str("id"));
            if (it != bsdf_ids_.end())
                id_bsdf = it->second;
        }
        mcsd::Instance in;
        in.id_bsdf = id_bsdf;
        in.flip_normals = ReadBool(n, {"flip_normals", "flipNormals"}, false);
        const Mat4f to_world = ReadTransform(n.Child("transform"));
        std::memcpy(in.to_world, to_world.m, sizeof(in.to_world));
        const std::string type = n.Str("type");
        if (type == "cube")
            in.type = MCSD_INST_CUBE;
        else if (type == "rectangle")
            in.type = MCSD_INST_RECTANGLE;
        else if (type == "sphere")
        {
            in.type = MCSD_INST_SPHERE;
            const Node *radius = n.Child("float");
            in.sphere_radius = radius ? radius->Float("value", 1.0f) : 1.0f;
            const V3 c = ReadVec3(n, {"center"}, V3{0, 0, 0});
            in.sphere_center[0] = c.x, in.sphere_center[1] = c.y, in.sphere_center[2] = c.z;
        }
        else if (type == "disk")
            in.type = MCSD_INST_DISK;
        else if (type == "cylinder")
        {
            in.type = MCSD_INST_CYLINDER;
            const V3 p0 = ReadVec3(n, {"p0"}, V3{0, 0, 0}), p1 = ReadVec3(n, {"p1"}, V3{0, 0, 1});
            in.cyl_p0[0] = p0.x, in.cyl_p0[1] = p0.y, in.cyl_p0[2] = p0.z;
            in.cyl_p1[0] = p1.x, in.cyl_p1[1] = p1.y, in.cyl_p1[2] = p1.z;
            const Node *radius = n.Child("float");
            in.cyl_radius = radius ? radius->Float("value", 1.0f) : 1.0f;
        }
        else if (type == "obj" || type == "serialized" || type == "ply" || type == "gltf")
        {
            in.type = MCSD_INST_MESHES;
            const Node *file = n.Child("string");
            const std::string name = file ? file->Str("value") : "";
            const std::string path = directory_ + name;
            const bool face_normals = ReadBool(n, {"face_normals", "faceNormals"}, false);
            MeshData mesh;
            // a file that is not there: the reference fails (model_loader.cpp:440-447) — and so does this
            // loader unless the caller supplied a stand-in for exactly this file (standin_mesh.cpp)
            if (standins_.Has(name) && !std::ifstream(path, std::ios::binary))
                mesh = standins_.Build(name);
            else if (type == "obj")
                mesh = LoadObj(path, ReadBool(n, {"flip_tex_coords", "flipTexCoords"}, true), face_normals);
            else if (type == "ply")
                mesh = LoadPly(path, face_normals);
            else if (type == "gltf")
                mesh = LoadGltf(path, face_normals); // (parser.cpp:1189: no texture-coordinate flip is requested for this type)
            else
            {
                const Node *index = n.Child("integer");
                mesh = LoadSerialized(path, index ? index->Int("value", 0) : 0);
            }
            in.positions = std::move(mesh.positions);
            in.normals = std::move(mesh.normals);
            in.texcoords = std::move(mesh.texcoords);
            in.tangents = std::move(mesh.tangents);
            in.bitangents = std::move(mesh.bitangents);
            in.indices = std::move(mesh.indices);
        }
        else
            throw std::runtime_error("unsupported shape type '" + type + "'.");
        if (const Node *m = ChildByName(n, {"interior"}))
            in.id_medium_int = ReadMedium(*m);
        if (const Node *m = ChildByName(n, {"exterior"}))
            in.id_medium_ext = ReadMedium(*m);
        out_.instances.push_back(std::move(in));
    }

    // ---- emitters (parser.cpp:1224-1428) -----------------------------------------------
    void ReadEmitter(const Node &n)
    {
        const std::string type = n.Str("type");
        mcsd::Emitter e;
        auto store3 = [](float *dst, V3 v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; };
        if (type == "point")
        {
            e.type = MCSD_EMIT_POINT;
            const Mat4f to_world = (ReadTransform(n.Child("transform")));
            store3(e.position, transform_point(to_world, ReadVec3(n, {"position"}, V3{0, 0, 0})));
            store3(e.intensity, ReadVec3(n, {"intensity"}, V3{1, 1, 1}));
        }
        else if (type == "spot")
        {
            e.type = MCSD_EMIT_SPOT;
            store3(e.intensity, ReadVec3(n, {"intensity"}, V3{1, 1, 1}));
            const Mat4f to_world = ReadTransform(n.Child("transform"));
            std::memcpy(e.to_world, to_world.m, sizeof(e.to_world));
            const float cutoff = ReadFloat(n, {"cutoff_angle", "cutoffAngle"}, 20);
            const float beam = ReadFloat(n, {"beamWidth", "beam_width"}, cutoff * 0.75f);
            e.cutoff_angle = Radians(cutoff), e.beam_width = Radians(beam);
            e.id_texture = kNone;
            if (const Node *t = n.Child("texture"))
                e.id_texture = ReadTexture(t, 1.0f, 1.0f);
        }
        else if (type == "directional")
        {
            e.type = MCSD_EMIT_DIRECTIONAL;
            const Mat4f to_world = ReadTransform(n.Child("transform"));
            const V3 local = ReadVec3(n, {"direction"}, V3{0, 0, 1});
            store3(e.direction, transform_dir((Inverted(Transposed(to_world))), local));
            store3(e.radiance, ReadVec3(n, {"radiance", "irradiance"}, V3{1, 1, 1}));
        }
        else if (type == "envmap")
        {
            e.type = MCSD_EMIT_ENVMAP;
            const Node *file = n.Child("string");
            const std::string filename = file ? file->Str("value") : "";
            const float gamma = ReadFloat(n, {"gamma"}, 0.0f), scale = ReadFloat(n, {"scale"}, 1.0f);
            const int width_target = static_cast<int>(out_.camera.width * 360 / out_.camera.fov_x);
            e.id_radiance = AddBitmap(directory_ + filename, filename, gamma, scale, &width_target);
            const Mat4f to_world = ReadTransform(n.Child("transform"));
            std::memcpy(e.to_world, to_world.m, sizeof(e.to_world));
        }
        else if (type == "constant")
        {
            e.type = MCSD_EMIT_CONSTANT;
            store3(e.radiance, ReadVec3(n, {"radiance"}, V3{1, 1, 1}));
        }
        else if (type == "sun" || type == "sky" || type == "sunsky")
            throw std::runtime_error("emitter '" + type + "' (Hosek-Wilkie sun / sky model) is not supported by this front end.");
        else
        {
            std::fprintf(stderr, "[warning] unsupport emitter '%s', ignore it.\n", type.c_str());
            return;
        }
        out_.emitters.push_back(e);
    }

    std::string directory_;
    StandinTable standins_;
    mcsd::Scene out_;
    std::map<std::string, std::string> defaults_;
    std::map<std::string, uint32_t> texture_ids_, bsdf_ids_, medium_ids_;
};

} // namespace

mcsd::Scene LoadXmlScene(const std::string &path, const std::string &standins)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw std::runtime_error("cannot find config file: '" + path + ".");
    const size_t dot = path.find_last_of('.');
    if (dot == std::string::npos || path.substr(dot + 1) != "xml")
        throw std::runtime_error("[error] only support mitsuba xml format config file.");
    std::stringstream buffer;
    buffer << f.rdbuf();
    const std::unique_ptr<xml::Node> root = xml::Parse(buffer.str());
    if (root->name != "scene")
        throw std::runtime_error("[error] read config file failed.");
    return SceneBuilder(path, StandinTable(standins)).Run(*root);
}

} // namespace mcpt
