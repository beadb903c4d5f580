// Flat scene layout shared by the host commit (csrc/host/commit.cpp) and the
// HIP kernels (csrc/hip/*.hip).  Everything the renderer reads on the GPU is
// a plain array in HBM addressed by index; there are no pointers inside
// records.  This replaces the reference's pointer graph of managed-memory
// objects (Scene/TLAS/BLAS/Instance/Primitive/Bsdf/Texture/Emitter/Medium,
// reference include/csrt/**) committed by Renderer::Renderer
// (src/renderer/renderer.cpp:259-348).
//
// Geometry layout (what the traversal kernel streams):
//   nodes      2 x float4 per BVH node (32 B):
//                [0] = box.lo.xyz, bits(skip)
//                [1] = box.hi.xyz, bits(object)   object = kNoObject for inner nodes
//              All trees live in one array, [TLAS | BLAS 0 | BLAS 1 | ...]
//              (reference scene.cpp:497-508), each tree numbered in PRE-ORDER
//              exactly as the reference builder numbers it
//              (bvh_builder.cpp:143-170), so `left child = index + 1` and the
//              fixed left-first visiting order of the reference
//              (blas.cpp:26-43, tlas.cpp:22-41) is reproduced WITHOUT a stack
//              by following `skip` (the pre-order successor that leaves the
//              subtree) on a miss or after a leaf.  kEndOfTree ends a tree.
//              The right child of node i is skip(i + 1).
//   walk_nodes 4 x float4 per node (64 B) of the ORDERED-walk hierarchy: a binary
//              SAH tree over all primitives of all instances (one primitive per
//              leaf, leaves have no record of their own):
//                [0] = child0.lo.xyz, bits(ref0)   [1] = child0.hi.xyz, bits(ref1)
//                [2] = child1.lo.xyz, -            [3] = child1.hi.xyz, -
//              ref = node index, or kWalkLeaf | slot into walk_prims.  Node 0 is
//              a top node whose child 0 is the whole scene (so the scene box is
//              tested like the reference's TLAS root) and child 1 is empty.
//              The boxes of leaves are bit-identical to the reference's leaf
//              boxes, interior boxes are exact unions: a primitive is reachable
//              here under exactly the conditions it is in the reference's tree.
//   wide_nodes 4 x uint4 per node (64 B) of the SAME hierarchy collapsed to four children per node, child boxes as
//              8-bit offsets on the node's own grid (commit.cpp, BuildWideNodes):
//                [0] = origin.xyz (float bits), biased exponents ex | ey << 8 | ez << 16 | n_children << 24
//                [1] = child references 0..3 (node index, kWalkLeaf | slot, or kWalkDone for an unused slot)
//                [2] = lo.x of children 0..3 (one byte each), lo.y, lo.z, hi.x     [3] = hi.y, hi.z, -, -
//              plane = origin + 2^(e - 127) * q: decoded boxes contain the exact ones (verified with these very
//              operations by the quantiser), so the walk reaches every primitive the exact hierarchy reaches; the exact
//              leaf-box test at the primitive (test_slot, kLeafCheck) stops what the larger boxes let through.
//   pool_nodes 8 x float4 per node (128 B) of the SAME hierarchy with four children per node and exact boxes (small scenes only;
//              the layout is at DeviceScene::pool_nodes)
//   walk_prims 3 x float4 per slot: p0 | bits(global primitive), p1 | bits(instance),
//              p2 | bits(rank of the primitive in the reference's visiting order)
//   tri_pos    3 x float4 per triangle: p0, p1, p2 (w unused)  -> 36 B useful
//   tri_attr   9 x float4 per triangle: n0 n1 n2 | t0 t1 t2 | b0 b1 b2, the six
//              texture coordinates ride in the w lanes (n0.w=u0 n1.w=v0
//              n2.w=u1 t0.w=v1 t1.w=u2 t2.w=v2) -> 132 B useful, read once per
//              shaded hit.
#ifndef MCPT_DEVICE_SCENE_H
#define MCPT_DEVICE_SCENE_H

#include <stdint.h>

#include "wave_target.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
struct float4
{
    float x, y, z, w;
};
struct uint4
{
    uint32_t x, y, z, w;
};
#endif

namespace mcpt
{

constexpr uint32_t kNoObject = 0xFFFFFFFFu;
constexpr uint32_t kEndOfTree = 0xFFFFFFFFu;
constexpr uint32_t kNone = 0xFFFFFFFFu; // reference kInvalidId (defs.hpp:22)
constexpr uint32_t kWalkLeaf = 0x80000000u; // walk_nodes child reference: primitive slot, not a node
constexpr uint32_t kWalkDone = 0xFFFFFFFFu; // bottom-of-stack sentinel of the ordered walk
constexpr uint32_t kWalkSliver = 0x80000000u; // walk_prims rank word: the triangle's computed distance is unreliable
                                              // (commit.cpp); the low 31 bits are the rank
constexpr uint32_t kWalkDepthMax = 56;      // bound on the ordered-walk tree depth
constexpr uint32_t kWalkStackMax = kWalkDepthMax + 1; // stack entries: one per level + the sentinel
constexpr uint32_t kWideRing = 16;      // entries per lane of the 4-wide walk's short stack that live in LDS (short_stack.h)
constexpr int kLutRes = 128;            // kulla_conty.hpp:9

struct Vec3f
{
    float x, y, z;
};

// Row-major 4x4.
struct Mat4f
{
    float m[16];
};

enum InstanceKind : uint32_t
{
    kInstTriangles = 0,
    kInstSphere = 1,
    kInstDisk = 2,
    kInstCylinder = 3,
};

// One record per instance.  blas_root and prim_base are GLOBAL indices.
struct InstanceRec
{
    uint32_t blas_root;
    uint32_t prim_base;
    uint32_t kind;     // InstanceKind
    uint32_t analytic; // index into analytic[] for non-triangle kinds
    uint32_t bsdf;     // kNone = no BSDF (pass-through surface)
    uint32_t medium_int, medium_ext;
    uint32_t area_light; // index of the area light this instance is, or kNone
    float pdf_area;      // 1 / instance "area" (reference scene.cpp:493-495)
    uint32_t pad[3];
};

// Analytic quadric with the matrices the reference recomputes on every test
// (sphere.cpp:21,47 etc.) precomputed once with the same inverse arithmetic.
struct AnalyticRec
{
    float radius, length;
    Vec3f center;
    float pad[3];
    Mat4f to_world, to_local, normal_to_world;
};

enum TextureKind : uint32_t
{
    kTexConstant = 1,
    kTexChecker = 2,
    kTexBitmap = 3,
};

struct TextureRec
{
    uint32_t kind;
    int32_t width, height, channel;
    Vec3f color;  // constant
    Vec3f color0; // checkerboard
    Vec3f color1;
    uint32_t texel_base; // offset into texels[]
    uint32_t pad[2];
    Mat4f to_uv;
};

enum BsdfKind : uint32_t
{
    kBsdfAreaLight = 1,
    kBsdfDiffuse = 2,
    kBsdfRoughDiffuse = 3,
    kBsdfConductor = 4,
    kBsdfDielectric = 5,
    kBsdfThinDielectric = 6,
    kBsdfPlastic = 7,
};

struct BsdfRec
{
    uint32_t kind;
    uint32_t twosided;
    uint32_t opacity, bump;       // texture ids or kNone
    uint32_t tex0, tex1, tex2, tex3;
    // tex0..3 by kind:
    //   area light     radiance
    //   diffuse        reflectance
    //   rough diffuse  reflectance, roughness
    //   conductor      roughness_u, roughness_v, specular_reflectance
    //   (thin) dielec. roughness_u, roughness_v, specular_reflectance, specular_transmittance
    //   plastic        roughness, diffuse_reflectance, specular_reflectance
    Vec3f reflectivity3; // conductor
    Vec3f f_avg3;        // conductor: hemispherical average Fresnel
    float reflectivity;  // dielectric / plastic: ((eta-1)/(eta+1))^2
    float eta, eta_inv;
    float f_avg, f_avg_inv;
    // dielectric: the transmitted share of the multiple-scattering term (dielectric.cpp:30-33 evaluates it per call, in double, from
    // the four constants above: a function of the record and of the side alone) seen from outside / from inside
    float ms_ratio_t, ms_ratio_t_inside;
};

struct MediumRec
{
    float sampling_weight;
    Vec3f sigma_s, sigma_t;
    uint32_t hg; // 1 = Henyey-Greenstein, 0 = isotropic
    Vec3f g;
    uint32_t pad;
};

enum EmitterKind : uint32_t
{
    kEmitPoint = 1,
    kEmitSpot = 2,
    kEmitDirectional = 3,
    kEmitSun = 4,
    kEmitEnvMap = 5,
    kEmitConstant = 6,
};

struct EmitterRec
{
    uint32_t kind;
    uint32_t texture; // spot / sun / envmap
    float cutoff, cos_cutoff, uv_factor, cos_beam, transition_rcp;
    float normalization; // envmap
    Vec3f position, intensity, direction, radiance;
    int32_t width, height; // envmap
    uint32_t cdf_cols, cdf_rows, weight_rows; // offsets into env_tables[] (Q7 pointer quirk)
    uint32_t pad[3];
    Mat4f to_world, to_local;
};

struct CameraRec // reference camera.cpp:26-37
{
    int32_t width, height;
    uint32_t spp;
    float spp_inv;
    Vec3f eye, front, dx, dy;
};

// (experiment builds: DeviceScene::wide_nodes holds PAIRS of quantised records, 128 bytes per node item — host/commit.cpp BuildWidePairs,
//  pool_walk.h; a property of the whole library, host and device side)
#ifndef MCPT_POOL_PAIRS
#define MCPT_POOL_PAIRS 0
#endif

struct IntegratorRec // reference integrator.hpp:31-69 (the scalar part)
{
    uint32_t volpath, hide_emitters;
    float pdf_rr, rr_scale; // rr_scale == pdf_rr (renderer.cpp:634, quirk Q2)
    uint32_t depth_rr, depth_max;
    uint32_t n_emitters, n_area_lights;
    uint32_t id_sun, id_envmap;
    uint32_t n_tlas_nodes, n_nodes, n_instances, n_prims;
    uint32_t n_walk_nodes; // ordered-walk hierarchy (0 = empty scene)
    uint32_t n_wide_nodes; // ... its 4-wide quantised form (wide_nodes)
    uint32_t wide_stack;   // stack entries a walk of the wide form can need (sentinel included)
    uint32_t walk_depth;   // stack entries a lane needs: the tree's depth + 1 (sentinel)
    uint32_t has_masks;    // some BSDF carries an opacity map: the walk must keep the reference's order
    uint32_t has_transmission; // some BSDF is a dielectric or a thin dielectric
    uint32_t has_non_conductor; // some BSDF is a rough diffuse, dielectric, thin dielectric or plastic model
    uint32_t has_reflectors;   // some BSDF is a rough diffuse, conductor, thin dielectric or plastic model
    uint32_t walk_hold;    // ... or when at least this many lanes hold a primitive (0 = never for that reason)
    uint32_t walk_break;   // wavefront scheduling of the ordered walk (traversal.h, walk_ordered_vote): leave
                           // the node phase when fewer lanes than this are searching; 0 = wait for all
    float walk_tie;        // distance below which two hits of the ordered walk count as tied (traversal.h,
                           // test_slot): ~40 float roundings at the scene's largest coordinate
    float walk_extent;     // largest |coordinate| of any box plane of the ordered-walk hierarchy (fused slab test: traversal.h)
    float walk_sliver_reach; // culling slack while the best hit is a sliver (kWalkSliver): the largest amount a
                             // sliver's leaf box was grown by (commit.cpp), never below walk_tie; 0 = the scene has none
    uint32_t n_pool_nodes;   // 4-wide exact form of the ordered-walk hierarchy (pool_nodes); 0 = not built (large scenes)
    uint32_t pool_depth;     // ... its depth
};

// Feature bits: which parts of the hot path a scene actually exercises.  The
// launcher picks a kernel instantiation from them.
enum SceneFeature : uint32_t
{
    kFeatVolPath = 1u << 0,     // volpath integrator / media
    kFeatEmitters = 1u << 1,    // non-area emitters present
    kFeatAnalytic = 1u << 2,    // sphere / disk / cylinder instances
    kFeatTextures = 1u << 3,    // non-constant textures, bump or opacity maps
    kFeatMicrofacet = 1u << 4,  // any BSDF other than diffuse / area light
    // not a property of the scene but of the launch: walk the SAH hierarchy
    // near child first instead of the reference's tree in the reference's order
    kFeatOrderedWalk = 1u << 5,
    // the ordered walk schedules its phases by wavefront vote (large scenes)
    kFeatVoteWalk = 1u << 6,
    // the scene has sliver triangles (IntegratorRec::walk_sliver_reach > 0): test_slot handles them
    kFeatSlivers = 1u << 7,
    // the ordered walk runs on the 4-wide quantised hierarchy (wide_nodes) with a short stack (short_stack.h)
    kFeatWideWalk = 1u << 8,
    // stream kernel only: the register budget of the instantiation — 3 / 2 wavefronts per SIMD instead of 4 (168 / 256 VGPRs:
    // 210 / 4 spilled registers instead of 288); which one is faster depends on the scene (stream_kernel_impl.h, StreamBudget)
    kFeatWaves3 = 1u << 9,
    kFeatWaves2 = 1u << 10,
    // stream kernel only: the instantiation leaves the transmissive BSDFs (dielectric, thin dielectric) out — for scenes
    // without one (IntegratorRec::has_transmission == 0): 210 -> 53 spilled VGPRs at the 3-wavefront budget
    kFeatNoTransmission = 1u << 11,
    // stream kernel only: of the models beyond diffuse only the dielectric is compiled in — for scenes whose other BSDFs are all
    // diffuse (IntegratorRec::has_reflectors == 0: matpreview rough dielectric): 288 -> 214 spilled VGPRs at the default budget
    kFeatDielectricOnly = 1u << 12,
    // ... or only the conductor (IntegratorRec::has_non_conductor == 0: matpreview rough conductor): 79 -> 53 spilled VGPRs at 3
    kFeatConductorOnly = 1u << 13,
    // LDS-resident lane kernels only: the ray queries are the wavefront-cooperative pool walk (pool_walk.h) instead of one
    // walk per lane (scenes without slivers whose node and slot indices fit 10 bits)
    kFeatPoolWalk = 1u << 14,
    // ... with 32-bit items (node / slot indices up to 2^26) on a hierarchy that is read through the caches: scenes outside LDS
    kFeatPoolBig = 1u << 15,
    // names the instantiations of the LOW-DISCREPANCY build (hip/render_variants_lowdisc.hip, compiled with MCPT_LOW_DISCREPANCY:
    // what a draw of the random stream returns is decided by that macro in vecmath.h, for the whole translation unit)
    kFeatLowDisc = 1u << 16,
    // pool-walk kernels of the path integrator: MERGED QUERIES — a vertex's last shadow query travels with the next segment's
    // closest query, two ray records per lane (path_core.h, path_step_merged; pool_walk.h, kDual)
    kFeatPoolMerge = 1u << 17,
    // the kernel's workgroups are 128 lanes (the class-sorted kernels, hip/sorted_body.h): the per-lane traversal stacks in LDS
    // interleave at that stride instead of 256
    kFeatGroup128 = 1u << 18,
};
constexpr uint32_t kLowDiscMaxSpp = 8192; // (a Sobol point's sample index has 13 bits: vecmath.h ld_pack)

// Device view: raw pointers into HBM + the scalar records.
struct DeviceScene
{
    CameraRec camera;
    IntegratorRec integrator;
    uint32_t features;
    uint32_t pad0;

    const float4 *nodes;       // 2 per node
    const float *node_area;    // 1 per node
    const float4 *walk_nodes;  // 4 per node of the ordered-walk hierarchy
    const float4 *walk_prims;  // 3 per slot
    const uint4 *wide_nodes;   // 4 per node of the 4-wide quantised hierarchy (meshes: scenes outside LDS)
    uint32_t *walk_spill;      // backing store of the short traversal stacks (short_stack.h): wide_stack x lanes of a launch
    uint32_t walk_spill_lanes; // ... lanes it has room for
    const float4 *tri_pos;     // 3 per triangle slot (global primitive index)
    const float4 *tri_attr;    // 9 per triangle slot
    const InstanceRec *instances;
    const AnalyticRec *analytic;
    const uint32_t *light_inst; // area light k -> instance
    const float *light_cdf;     // n_area_lights + 1, not normalised
    const TextureRec *textures;
    const float *texels;
    const BsdfRec *bsdfs;
    const MediumRec *media;
    const EmitterRec *emitters;
    const float *env_tables;
    const float *lut_brdf;   // kLutRes * kLutRes
    const float *lut_albedo; // kLutRes
    // Primary-visibility pre-pass (hip/primary_kernel.hip), or null: the closest hit of the CAMERA ray of sample s of
    // pixel p, two words at 2 (item(p) * spp + s): primitive (kNone: miss) and instance.  The camera ray of a sample is a
    // function of (pixel, sample index) only — stratified in x, van der Corput in y, no random number
    // (renderer.cpp:68-76) — so all of a frame's camera rays can be traced ahead of the per-pixel sample chains, by
    // a lean kernel with coherent wavefronts, and the chains start every sample at its first vertex.
    const uint32_t *prehit;
    uint32_t prehit_step; // start_sample's step of the launch that reads prehit (split samples; 1 otherwise)
    // the tile enumeration of the draw the pre-pass was made for (RenderJob: tile_first, tile_stride, tiles_x): the buffer
    // holds the draw's OWN items only — record of (pixel, sample) at 2 (item(pixel) * spp + sample), path_core.h::prehit_record
    uint32_t prehit_tile_first, prehit_tile_stride, prehit_tiles_x;
    // The ordered-walk hierarchy collapsed to FOUR children per node with their exact boxes, for the wavefront-cooperative
    // pool walk of small scenes (pool_walk.h; commit.cpp, BuildPoolNodes): 8 x float4 = 128 B per node, plane-major so that
    // the walk READS the planes a ray enters / leaves through instead of selecting them:
    //   [0] lo.x of children 0..3   [1] lo.y   [2] lo.z   [3] hi.x   [4] hi.y   [5] hi.z   [6] the four references   [7] -
    // reference = node index or kWalkLeaf | slot; an unused child has an inverted box (never entered) and names slot 0.
    const float4 *pool_nodes;
};

// Counters of the measurement mode: totals over a launch.
struct TraceCounters
{
    unsigned long long closest_rays, shadow_rays, node_tests, prim_tests,
        shaded_hits, samples, wave_node_steps, wave_prim_steps;
    unsigned long long ticks_shade, ticks_trace, ticks_wait, rounds; // stream kernel: where its wavefronts spend their time
};

} // namespace mcpt

#endif // MCPT_DEVICE_SCENE_H
