// Built-in scenes: the renderer configuration the XML front end produces for
// the reference's example scenes, assembled in code so that the GPU path can
// be exercised without any scene file.
#include <cstring>
#include <stdexcept>

#include "../vecmath.h"
#include "frontend.hpp"

namespace mcpt
{
namespace
{

uint32_t AddConstantTexture(mcsd::Scene &s, float r, float g, float b)
{
    mcsd::Texture t;
    t.type = MCSD_TEX_CONSTANT;
    t.color[0] = r, t.color[1] = g, t.color[2] = b;
    s.textures.push_back(t);
    return static_cast<uint32_t>(s.textures.size() - 1);
}

// A two-sided Lambertian BSDF with its own constant reflectance texture, in
// the order the reference parser creates them (parser.cpp:651-671, 834-847).
uint32_t AddDiffuse(mcsd::Scene &s, float r, float g, float b)
{
    mcsd::Bsdf bsdf;
    bsdf.type = MCSD_BSDF_DIFFUSE;
    bsdf.twosided = 1;
    bsdf.id_diffuse_reflectance = AddConstantTexture(s, r, g, b);
    s.bsdfs.push_back(bsdf);
    return static_cast<uint32_t>(s.bsdfs.size() - 1);
}

void AddShape(mcsd::Scene &s, uint32_t type, uint32_t bsdf, const float (&m)[16])
{
    mcsd::Instance in;
    in.type = type;
    in.id_bsdf = bsdf;
    std::memcpy(in.to_world, m, sizeof(m));
    s.instances.push_back(in);
}

// resources/scene/cornell-box/scene_v0.6.xml of the reference: five walls, two
// boxes, one rectangular area light (radiance 17, 12, 4), path integrator with
// maxDepth 65, Russian roulette from depth 5 at 0.95.
mcsd::Scene CornellBox()
{
    mcsd::Scene s;
    s.camera.spp = 256;
    s.camera.width = 1024, s.camera.height = 1024;
    s.camera.fov_x = 19.5f;
    // sensor toWorld = [-1 0 0 0; 0 1 0 1; 0 0 -1 6.8; 0 0 0 1] applied to the
    // origin, +z and +y (parser.cpp:345-356)
    const Mat4f to_world = {{-1, 0, 0, 0, 0, 1, 0, 1, 0, 0, -1, 6.8f, 0, 0, 0, 1}};
    const V3 eye = transform_point(to_world, V3{0, 0, 0}), look_at = transform_point(to_world, V3{0, 0, 1}),
             up = transform_dir(to_world, V3{0, 1, 0});
    s.camera.eye[0] = eye.x, s.camera.eye[1] = eye.y, s.camera.eye[2] = eye.z;
    s.camera.look_at[0] = look_at.x, s.camera.look_at[1] = look_at.y, s.camera.look_at[2] = look_at.z;
    s.camera.up[0] = up.x, s.camera.up[1] = up.y, s.camera.up[2] = up.z;
    s.integrator.type = MCSD_INTEGRATOR_PATH;
    s.integrator.depth_max = 65, s.integrator.depth_rr = 5, s.integrator.pdf_rr = 0.95f;

    const uint32_t left = AddDiffuse(s, 0.63f, 0.065f, 0.05f), right = AddDiffuse(s, 0.14f, 0.45f, 0.091f),
                   floor = AddDiffuse(s, 0.725f, 0.71f, 0.68f), ceiling = AddDiffuse(s, 0.725f, 0.71f, 0.68f),
                   back = AddDiffuse(s, 0.725f, 0.71f, 0.68f), short_box = AddDiffuse(s, 0.725f, 0.71f, 0.68f),
                   tall_box = AddDiffuse(s, 0.725f, 0.71f, 0.68f);
    AddDiffuse(s, 0, 0, 0); // "Light" BSDF: declared, superseded by the emitter

    AddShape(s, MCSD_INST_RECTANGLE, floor, {0, 1, 0, 0, 0, 0, 2, 0, 1, 0, 0, 0, 0, 0, 0, 1});
    AddShape(s, MCSD_INST_RECTANGLE, ceiling, {-1, 0, 0, 0, 0, 0, -2, 2, 0, -1, 0, 0, 0, 0, 0, 1});
    AddShape(s, MCSD_INST_RECTANGLE, back, {0, 1, 0, 0, 1, 0, 0, 1, 0, 0, -2, -1, 0, 0, 0, 1});
    AddShape(s, MCSD_INST_RECTANGLE, right, {0, 0, 2, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 1});
    AddShape(s, MCSD_INST_RECTANGLE, left, {0, 0, -2, -1, 1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 0, 1});
    AddShape(s, MCSD_INST_CUBE, short_box,
             {0.0851643f, 0.289542f, 1.31134e-008f, 0.328631f, 3.72265e-009f, 1.26563e-008f, -0.3f, 0.3f, -0.284951f,
              0.0865363f, 5.73206e-016f, 0.374592f, 0, 0, 0, 1});
    AddShape(s, MCSD_INST_CUBE, tall_box,
             {0.286776f, 0.098229f, -2.29282e-015f, -0.335439f, -4.36233e-009f, 1.23382e-008f, -0.6f, 0.6f,
              -0.0997984f, 0.282266f, 2.62268e-008f, -0.291415f, 0, 0, 0, 1});

    // a shape with an <emitter> child gets a pseudo-BSDF of kind "area light"
    // (parser.cpp:1068-1100)
    mcsd::Bsdf light;
    light.type = MCSD_BSDF_AREA_LIGHT;
    light.twosided = 0;
    light.weight = 1.0f;
    light.id_radiance = AddConstantTexture(s, 17, 12, 4);
    s.bsdfs.push_back(light);
    AddShape(s, MCSD_INST_RECTANGLE, static_cast<uint32_t>(s.bsdfs.size() - 1),
             {0.235f, 0, 0, -0.005f, 0, 0, -0.0893f, 1.98f, 0, 0.19f, 0, -0.03f, 0, 0, 0, 1});
    return s;
}

} // namespace

mcsd::Scene BuiltinScene(const std::string &name)
{
    if (name == "cornell-box" || name == "cornell_box" || name == "cornell")
        return CornellBox();
    throw std::runtime_error("unknown built-in scene '" + name + "'.");
}

} // namespace mcpt
