// Host-side scene commit: mcsd::Scene (the RendererConfig equivalent) ->
// FlatScene (plain arrays in the layout of device_scene.h).  This is the work
// the reference does in Renderer::Renderer / Scene::Scene
// (reference src/renderer/renderer.cpp:259-348, src/rtcore/scene.cpp:118-533).
#ifndef MCPT_HOST_COMMIT_HPP
#define MCPT_HOST_COMMIT_HPP

#include <vector>

#include "../device_scene.h"
#include "mcsd_scene.hpp"

namespace mcpt
{

struct FlatScene
{
    CameraRec camera{};
    IntegratorRec integrator{};
    uint32_t features = 0;

    std::vector<float4> nodes;     // 2 per node
    std::vector<float> node_area;  // 1 per node
    std::vector<float4> walk_nodes; // 4 per node (ordered-walk hierarchy)
    std::vector<float4> walk_prims; // 3 per slot
    std::vector<uint4> wide_nodes;  // 4 per node of the 4-wide quantised form of the ordered-walk hierarchy
    std::vector<float4> pool_nodes; // 8 per node of its 4-wide EXACT form
    std::vector<float4> tri_pos;   // 3 per primitive slot
    std::vector<float4> tri_attr;  // 9 per primitive slot
    std::vector<InstanceRec> instances;
    std::vector<AnalyticRec> analytic;
    std::vector<uint32_t> light_inst;
    std::vector<float> light_cdf;
    std::vector<TextureRec> textures;
    std::vector<float> texels;
    std::vector<BsdfRec> bsdfs;
    std::vector<MediumRec> media;
    std::vector<EmitterRec> emitters;
    std::vector<float> env_tables;
    std::vector<float> lut_brdf, lut_albedo;

    // host seconds spent in the commit: baking + reference-topology LBVHs, the
    // ordered-walk hierarchy, everything
    double seconds_lbvh = 0, seconds_walk = 0, seconds_total = 0;

    // A DeviceScene whose pointers address the host vectors (CPU inspection,
    // tests) — the uploader rewrites them to HBM addresses.
    DeviceScene HostView() const;
    size_t GeometryBytes() const;
};

// Optional accelerator for the reference-topology LBVH of large instances: same
// contract as BuildReferenceLbvh (bit-identical output), e.g. the HIP builder of
// hip/lbvh_build.hip.  Instances with fewer than `min_prims` primitives are built on
// the host.
struct LbvhAccelerator
{
    virtual ~LbvhAccelerator() = default;
    uint32_t min_prims = 1u << 16;
    virtual void Build(uint32_t n, const float *boxes, const float *areas, std::vector<float4> &nodes,
                       std::vector<float> &node_area) = 0;
};

// TESTS ONLY: how the ordered-walk hierarchy is split.  0 = binned SAH (production),
// 1 = exact SAH sweep, 2 = object median, 3 = binned SAH with the children swapped.  The
// image must not depend on it (tests/test_host.py::test_image_does_not_depend_on_the_walk_tree).
void SetWalkTreeStrategyForTesting(int strategy);
void SetWalkTieScaleForTesting(float scale); // 1 = production; tests shrink the ordered walk's tie radius with it

// Throws std::runtime_error with the reference's wording on invalid input.
FlatScene CommitScene(const mcsd::Scene &scene, LbvhAccelerator *lbvh = nullptr);

// The reference-topology LBVH of one set of boxes (bvh_builder.cpp:74-207) in the
// layout of device_scene.h: 2 float4 + 1 area per node, 2n-1 nodes, tree-local
// links, object = box index.  boxes: 6 floats each (lo.xyz, hi.xyz).
void BuildReferenceLbvh(uint32_t n, const float *boxes, const float *areas, std::vector<float4> &nodes,
                        std::vector<float> &node_area);

// The 128x128 + 128 Kulla-Conty tables (reference kulla_conty.cpp:62-80),
// computed once per process (multi-threaded, ~0.5 s) and cached.
void KullaContyTables(const float **brdf, const float **albedo);

} // namespace mcpt

#endif // MCPT_HOST_COMMIT_HPP
