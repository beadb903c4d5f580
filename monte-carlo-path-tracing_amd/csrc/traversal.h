// Ray / scene intersection on the flat scene of device_scene.h.
//
// Two ray queries with the same answers:
//   walk_ordered / walk_ordered_vote   the production query: near-child-first walk of
//                                      the SAH hierarchy (further down in this file);
//   walk_scene                         the reference's trees in the reference's order
//                                      (scenes with opacity masks; validation mode).
//
// * walk_scene — STACKLESS two-level traversal: nodes are stored in the reference
//   builder's pre-order, so "descend" is `index + 1` and every other move is
//   the node's precomputed `skip` link.  This visits nodes in exactly the
//   reference's order (left child first, then right; tlas.cpp:22-41,
//   blas.cpp:26-43) with one index register instead of two 65-entry stacks.
//   A TLAS leaf jumps to the instance's BLAS root and remembers the TLAS skip
//   link to resume from when the BLAS walk ends.
// * Closest hit keeps shrinking t_max; a candidate with t == t_max is accepted
//   (triangle.cpp:82: only t > t_max rejects), so among equal distances the
//   LAST visited primitive wins — as in the reference.
// * Triangles: Woop et al. watertight test with the double-precision edge
//   fallback (triangle.cpp:23-87, ray.cpp:26-45).  Only the barycentrics are
//   produced during the walk; the shading frame is interpolated once, for the
//   final hit (same arithmetic as triangle.cpp:122-144).
// * Quadrics (sphere/disk/cylinder.cpp) use matrices inverted once on the host
//   with the reference's inverse arithmetic instead of per test.
#ifndef MCPT_TRAVERSAL_H
#define MCPT_TRAVERSAL_H

#include "bsdfs.h"
#include "phase_clock.h"

namespace mcpt
{

struct Ray // ray.hpp:9-27
{
    V3 origin, dir, dir_rcp, shear;
    float t_max;
    int kx, ky, kz;
};

MCPT_HD Ray make_ray(V3 origin, V3 dir) // ray.cpp:18-47
{
    Ray r;
    r.origin = origin, r.dir = dir, r.t_max = kMaxFloat;
    r.dir_rcp = V3{1.0f / (dir.x != 0 ? dir.x : kEpsDistance), 1.0f / (dir.y != 0 ? dir.y : kEpsDistance),
                   1.0f / (dir.z != 0 ? dir.z : kEpsDistance)};
    const float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    r.kz = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
    r.kx = r.kz + 1 == 3 ? 0 : r.kz + 1;
    r.ky = r.kx + 1 == 3 ? 0 : r.kx + 1;
    const float dz = comp(dir, r.kz);
    if (dz < 0.0f)
    {
        const int t = r.kx;
        r.kx = r.ky;
        r.ky = t;
    }
    r.shear = V3{comp(dir, r.kx) / dz, comp(dir, r.ky) / dz, 1.0f / dz};
    return r;
}

// What the walk records about the closest primitive.  (a, b, c) are the
// barycentrics u, v, w of a triangle or the object-space hit point of a quadric.
struct HitRaw
{
    uint32_t inst, prim;
    float a, b, c;
    bool inside;
};

// Fully reconstructed surface point (hit.hpp:9-30).
struct Surface
{
    bool inside;
    uint32_t inst;
    V2 uv;
    V3 position, normal, tangent, bitangent;
};

struct TraceStats
{
    uint32_t node_tests, prim_tests;
    // ordered walk only: wavefront-level steps of the two phases, counted by one lane
    // per wavefront (lane utilisation = lane-level tests / (64 x these))
    uint32_t wave_node_steps, wave_prim_steps;
};

// aabb.cpp:29-48.  The reference computes (lo - o) * rcp and (hi - o) * rcp and
// then picks per axis by the sign of rcp; picking the plane first and
// transforming only the picked one is the same arithmetic on the same operands
// with half the subtract / multiply work.
MCPT_HD bool box_hit(const float4 &lo, const float4 &hi, const Ray &r)
{
    const bool px = r.dir_rcp.x > 0, py = r.dir_rcp.y > 0, pz = r.dir_rcp.z > 0;
    const float nx = ((px ? lo.x : hi.x) - r.origin.x) * r.dir_rcp.x, fx = ((px ? hi.x : lo.x) - r.origin.x) * r.dir_rcp.x;
    const float ny = ((py ? lo.y : hi.y) - r.origin.y) * r.dir_rcp.y, fy = ((py ? hi.y : lo.y) - r.origin.y) * r.dir_rcp.y;
    const float nz = ((pz ? lo.z : hi.z) - r.origin.z) * r.dir_rcp.z, fz = ((pz ? hi.z : lo.z) - r.origin.z) * r.dir_rcp.z;
    const float t_enter = fmaxf(fmaxf(fmaxf(kEpsDistance, nx), ny), nz);
    const float t_exit = fminf(fminf(fminf(r.t_max, fx), fy), fz);
    return t_enter <= t_exit;
}

MCPT_HD uint32_t as_uint(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
#endif
}

MCPT_HD float as_float(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

// Opacity mask of an instance's BSDF (bsdf.cpp:272-276): may draw one number.
template <bool kTextures>
MCPT_HD bool masked_out(const DeviceScene &sc, uint32_t bsdf, V2 uv, uint32_t &rng)
{
    if (!kTextures || bsdf == kNone)
        return false;
    const uint32_t opacity = sc.bsdfs[bsdf].opacity;
    return opacity != kNone && texture_transparent(sc.textures, sc.texels, opacity, uv, rng);
}

MCPT_HD V2 triangle_uv(const float4 *attr, float u, float v, float w) // Lerp(texcoords, u, v, w)
{
    const V2 uv0 = V2{attr[0].w, attr[1].w}, uv1 = V2{attr[2].w, attr[3].w}, uv2 = V2{attr[4].w, attr[5].w};
    return u * uv0 + v * uv1 + w * uv2;
}

// triangle.cpp:19-120 (WATERTIGHT_TRIANGLES branch)
template <bool kTextures>
MCPT_HD bool triangle_hit(const DeviceScene &sc, uint32_t prim, uint32_t bsdf, Ray &ray, uint32_t &rng, HitRaw &out)
{
    const float4 *p = sc.tri_pos + 3 * static_cast<size_t>(prim);
    const V3 A = xyz(p[0]) - ray.origin, B = xyz(p[1]) - ray.origin, C = xyz(p[2]) - ray.origin;
    const float Akz = comp(A, ray.kz), Bkz = comp(B, ray.kz), Ckz = comp(C, ray.kz);
    const float Ax = comp(A, ray.kx) - ray.shear.x * Akz, Ay = comp(A, ray.ky) - ray.shear.y * Akz;
    const float Bx = comp(B, ray.kx) - ray.shear.x * Bkz, By = comp(B, ray.ky) - ray.shear.y * Bkz;
    const float Cx = comp(C, ray.kx) - ray.shear.x * Ckz, Cy = comp(C, ray.ky) - ray.shear.y * Ckz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f)
    {
        U = static_cast<float>(D(Cx) * D(By) - D(Cy) * D(Bx));
        V = static_cast<float>(D(Ax) * D(Cy) - D(Ay) * D(Cx));
        W = static_cast<float>(D(Bx) * D(Ay) - D(By) * D(Ax));
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    const float det = U + V + W;
    if (det == 0.0f)
        return false;
    const float T = U * (ray.shear.z * Akz) + V * (ray.shear.z * Bkz) + W * (ray.shear.z * Ckz);
    const float det_inv = 1.0f / det;
    const float t = T * det_inv;
    if (t > ray.t_max || t < kEpsDistance)
        return false;
    const float u = U * det_inv, v = V * det_inv, w = W * det_inv;
    if (kTextures && masked_out<kTextures>(sc, bsdf, triangle_uv(sc.tri_attr + 9 * static_cast<size_t>(prim), u, v, w), rng))
        return false;
    ray.t_max = t;
    out.prim = prim, out.a = u, out.b = v, out.c = w, out.inside = det_inv < 0;
    return true;
}

MCPT_HD V2 sphere_uv(V3 p_local, float &theta, float &phi)
{
    to_spherical(p_local, theta, phi);
    return V2{phi * k1Div2Pi, theta * k1DivPi};
}

// sphere.cpp:17-46
template <bool kTextures>
MCPT_HD bool sphere_hit(const DeviceScene &sc, const AnalyticRec &q, uint32_t prim, uint32_t bsdf, Ray &ray,
                        uint32_t &rng, HitRaw &out)
{
    const V3 o = transform_point(q.to_local, ray.origin) - from(q.center), d = transform_dir(q.to_local, ray.dir);
    const float a = dot(d, d), b = 2.0f * dot(d, o), c = dot(o, o) - sqr(q.radius);
    float t_near = 0.0f, t_far = 0.0f;
    if (!solve_quadratic(a, b, c, t_near, t_far) || t_far < kEpsDistance)
        return false;
    float t = t_near < kEpsDistance ? t_far : t_near;
    const V3 p_local = o + t * d, position = transform_point(q.to_world, p_local + from(q.center));
    t = length(position - ray.origin);
    if (t > ray.t_max || t < kEpsDistance)
        return false;
    if (kTextures)
    {
        float theta, phi;
        if (masked_out<kTextures>(sc, bsdf, sphere_uv(p_local, theta, phi), rng))
            return false;
    }
    ray.t_max = t;
    out.prim = prim, out.a = p_local.x, out.b = p_local.y, out.c = p_local.z, out.inside = c < 0.0f;
    return true;
}

// disk.cpp:17-44
template <bool kTextures>
MCPT_HD bool disk_hit(const DeviceScene &sc, const AnalyticRec &q, uint32_t prim, uint32_t bsdf, Ray &ray,
                      uint32_t &rng, HitRaw &out)
{
    const V3 o = transform_point(q.to_local, ray.origin), d = transform_dir(q.to_local, ray.dir);
    const float t_z = -o.z / d.z;
    if (t_z < kEpsFloat)
        return false;
    const V3 p_local = o + t_z * d;
    if (length(p_local) > 0.5f)
        return false;
    const float t = length(transform_point(q.to_world, p_local) - ray.origin);
    if (t > ray.t_max || t < kEpsDistance)
        return false;
    if (kTextures)
    {
        float theta, phi;
        to_spherical(p_local, theta, phi);
        if (masked_out<kTextures>(sc, bsdf, V2{length(p_local), phi * k1Div2Pi}, rng))
            return false;
    }
    ray.t_max = t;
    out.prim = prim, out.a = p_local.x, out.b = p_local.y, out.c = p_local.z, out.inside = d.z > 0;
    return true;
}

// cylinder.cpp:21-59
template <bool kTextures>
MCPT_HD bool cylinder_hit(const DeviceScene &sc, const AnalyticRec &q, uint32_t prim, uint32_t bsdf, Ray &ray,
                          uint32_t &rng, HitRaw &out)
{
    const V3 o = transform_point(q.to_local, ray.origin), d = transform_dir(q.to_local, ray.dir);
    const float a = sqr(d.x) + sqr(d.y), b = 2.0f * (d.x * o.x + d.y * o.y), c = sqr(o.x) + sqr(o.y) - sqr(q.radius);
    float t_near = 0.0f, t_far = 0.0f;
    if (!solve_quadratic(a, b, c, t_near, t_far) || t_far < kEpsDistance)
        return false;
    const float z_near = o.z + d.z * t_near, z_far = o.z + d.z * t_far;
    float t;
    if (kEpsDistance < t_near && 0.0f <= z_near && z_near <= q.length)
        t = t_near;
    else if (0.0f <= z_far && z_far <= q.length)
        t = t_far;
    else
        return false;
    const V3 p_local = o + t * d;
    if (kTextures &&
        masked_out<kTextures>(sc, bsdf, V2{gl::atan2f(p_local.y, p_local.x) * k1Div2Pi, p_local.z / q.length}, rng))
        return false;
    t = length(transform_point(q.to_world, p_local) - ray.origin);
    if (t > ray.t_max || t < kEpsDistance)
        return false;
    ray.t_max = t;
    out.prim = prim, out.a = p_local.x, out.b = p_local.y, out.c = p_local.z, out.inside = c < 0.0f;
    return true;
}

// The walk.  kAny = shadow query: stop at the first accepted primitive.
// Returns whether anything was hit; for closest queries `hit` describes it and
// ray.t_max is its distance.
//
// Shape of the loop ("while-while"): an inner loop of pure node steps (one
// 32-byte node fetch + slab test each) runs until the lane holds a primitive
// candidate or has left the tree; only then is the (much longer) primitive
// test executed.  Per lane the visiting order is unchanged; per wavefront the
// 64 lanes run box tests together and primitive tests together instead of
// paying for both in every iteration.
template <bool kAny, bool kAnalytic, bool kTextures, bool kCount>
MCPT_HD bool walk_scene(const DeviceScene &sc, Ray &ray, uint32_t &rng, HitRaw &hit, TraceStats &stats)
{
    bool found = false;
    const float t_start = ray.t_max;
    uint32_t node = sc.integrator.n_tlas_nodes ? 0u : kEndOfTree;
    uint32_t resume = kEndOfTree; // TLAS link to continue from once the current BLAS is exhausted
    bool in_blas = false;
    uint32_t inst = 0, inst_kind = 0, inst_bsdf = kNone, inst_analytic = 0;
    for (;;)
    {
        // ---- node steps until a primitive candidate turns up ----------------
        uint32_t object = kNoObject, after = kEndOfTree;
        for (;;)
        {
            if (node == kEndOfTree)
            {
                if (!in_blas)
                    break;
                in_blas = false;
                node = resume;
                continue;
            }
            const float4 n0 = sc.nodes[2 * static_cast<size_t>(node)], n1 = sc.nodes[2 * static_cast<size_t>(node) + 1];
            if (kCount)
                ++stats.node_tests;
            const bool inside_box = box_hit(n0, n1, ray);
            const uint32_t skip = as_uint(n0.w), leaf_object = as_uint(n1.w);
            if (inside_box && leaf_object != kNoObject)
            {
                if (in_blas)
                {
                    object = leaf_object, after = skip;
                    break;
                }
                const InstanceRec &rec = sc.instances[leaf_object];
                inst = leaf_object, inst_kind = rec.kind, inst_bsdf = rec.bsdf, inst_analytic = rec.analytic;
                resume = skip;
                node = rec.blas_root;
                in_blas = true;
                continue;
            }
            // inner node: descend to the left child (pre-order successor) on a
            // hit, otherwise leave the subtree
            node = inside_box ? node + 1 : skip;
        }
        if (object == kNoObject)
            break; // left the TLAS: done

        // ---- primitive test ---------------------------------------------------
        if (kCount)
            ++stats.prim_tests;
        // Candidates are written to a local record and committed in ONE place:
        // with four writers merging into `hit` across divergent branches hipcc
        // -O2/-O3 (ROCm 7.2) dropped the disk path's store of `c` (found by the
        // host-vs-device differential test, tests/test_gpu_units.py).
        HitRaw cand;
        cand.inst = inst, cand.prim = object, cand.a = cand.b = cand.c = 0.0f, cand.inside = false;
        bool accepted;
        if (!kAnalytic || inst_kind == kInstTriangles)
            accepted = triangle_hit<kTextures>(sc, object, inst_bsdf, ray, rng, cand);
        else if (inst_kind == kInstSphere)
            accepted = sphere_hit<kTextures>(sc, sc.analytic[inst_analytic], object, inst_bsdf, ray, rng, cand);
        else if (inst_kind == kInstDisk)
            accepted = disk_hit<kTextures>(sc, sc.analytic[inst_analytic], object, inst_bsdf, ray, rng, cand);
        else
            accepted = cylinder_hit<kTextures>(sc, sc.analytic[inst_analytic], object, inst_bsdf, ray, rng, cand);
        if (accepted)
        {
            found = true;
            hit = cand;
            hit.inst = inst;
            if (kAny)
                return true;
        }
        node = after;
    }
    if (!kAny && found && ray.t_max != ray.t_max)
    {
        // NaN distance: a ray with NaN components passes every test (of the reference too), each
        // "hit" is accepted because `t > t_max` is false, and the result stays NaN.  The reference
        // walks an instance with a copy of the ray and merges the result only if
        // `local.t_max <= ray.t_max` (tlas.cpp:27-33), which drops it: the query misses.
        // (Decided once, here: a second writer of ray.t_max inside the loop made hipcc drop real
        //  sphere hits on gfx950 — tests/test_gpu_parity.py::test_ordered_walk_equals_reference_walk.)
        ray.t_max = t_start;
        found = false;
    }
    return found;
}


// ---- ordered walk -------------------------------------------------------------
// The production ray query: near-child-first traversal of the SAH hierarchy
// (walk_nodes / walk_prims, device_scene.h) with a small per-lane stack of
// postponed far children.  One step loads ONE 64-byte node and tests BOTH
// children, so a ray touches roughly half as many records as in the
// one-box-per-node walk above, and the shrinking t_max prunes most of the far
// side — the reference's fixed left-then-right order cannot do that.
//
// Same answers as walk_scene (see commit.cpp, WalkTreeBuilder, for why the
// same primitives are reachable):
//   * shadow queries: t_max is fixed, acceptance of a primitive does not depend
//     on the order, the result is the same boolean;
//   * closest queries: the minimum distance is order independent; among
//     primitives at EQUAL distance the reference keeps the one it visits last
//     (triangle.cpp:82 accepts t == t_max) — reproduced by comparing the
//     primitives' ranks in the reference's visiting order.
// Not usable when a BSDF carries an opacity map: that test draws random numbers
// during the walk, so the order of visits is part of the image
// (IntegratorRec::has_masks selects walk_scene then).
//
// Stack: `stack[level * kWalkStackStride]`; on the GPU the lanes of a workgroup
// interleave their stacks in LDS (stride = workgroup size, conflict free), the
// host build uses a plain array.
// (kernels with workgroups of 128 lanes — the class-sorted ones — say so with a feature bit: Config::kStackStride, path_core.h)
#if MCPT_WAVE_DEVICE
constexpr uint32_t kWalkStackStride = 256;
#else
constexpr uint32_t kWalkStackStride = 1;
#endif

// Slab test of box_hit, also returning the entry distance.
MCPT_HD bool box_enter(const float4 &lo, const float4 &hi, const Ray &r, float &t_enter)
{
    const bool px = r.dir_rcp.x > 0, py = r.dir_rcp.y > 0, pz = r.dir_rcp.z > 0;
    const float nx = ((px ? lo.x : hi.x) - r.origin.x) * r.dir_rcp.x, fx = ((px ? hi.x : lo.x) - r.origin.x) * r.dir_rcp.x;
    const float ny = ((py ? lo.y : hi.y) - r.origin.y) * r.dir_rcp.y, fy = ((py ? hi.y : lo.y) - r.origin.y) * r.dir_rcp.y;
    const float nz = ((pz ? lo.z : hi.z) - r.origin.z) * r.dir_rcp.z, fz = ((pz ? hi.z : lo.z) - r.origin.z) * r.dir_rcp.z;
    t_enter = fmaxf(fmaxf(fmaxf(kEpsDistance, nx), ny), nz);
    const float t_exit = fminf(fminf(fminf(r.t_max, fx), fy), fz);
    return t_enter <= t_exit;
}

// Watertight triangle test of triangle_hit on a walk_prims record, WITHOUT the
// distance bound (and without a mask): does the ray's line hit the triangle at
// t >= kEpsDistance, and where.  Predicated: one rare branch (the double-precision
// edge fallback), no early-outs.
struct SlotHit
{
    bool hit;
    float t, a, b, c;
    bool inside;
};

// (the test on the vertices' offsets from the origin, ALREADY in the ray's axis order: component kx, ky, kz of each)
MCPT_HD SlotHit triangle_probe_permuted(float Akx, float Aky, float Akz, float Bkx, float Bky, float Bkz, float Ckx, float Cky, float Ckz, V3 shear)
{
    const float Ax = Akx - shear.x * Akz, Ay = Aky - shear.y * Akz;
    const float Bx = Bkx - shear.x * Bkz, By = Bky - shear.y * Bkz;
    const float Cx = Ckx - shear.x * Ckz, Cy = Cky - shear.y * Ckz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f)
    {
        U = static_cast<float>(D(Cx) * D(By) - D(Cy) * D(Bx));
        V = static_cast<float>(D(Ax) * D(Cy) - D(Ay) * D(Cx));
        W = static_cast<float>(D(Bx) * D(Ay) - D(By) * D(Ax));
    }
    const bool mixed_signs = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
    const float det = U + V + W;
    const float T = U * (shear.z * Akz) + V * (shear.z * Bkz) + W * (shear.z * Ckz);
    const float det_inv = 1.0f / det;
    SlotHit h;
    h.t = T * det_inv;
    h.hit = !mixed_signs && det != 0.0f && !(h.t < kEpsDistance);
    h.a = U * det_inv, h.b = V * det_inv, h.c = W * det_inv, h.inside = det_inv < 0;
    return h;
}

MCPT_HD SlotHit triangle_probe(const float4 *p, const Ray &ray)
{
    const V3 A = xyz(p[0]) - ray.origin, B = xyz(p[1]) - ray.origin, C = xyz(p[2]) - ray.origin;
    return triangle_probe_permuted(comp(A, ray.kx), comp(A, ray.ky), comp(A, ray.kz), comp(B, ray.kx), comp(B, ray.ky), comp(B, ray.kz), comp(C, ray.kx),
                                   comp(C, ray.ky), comp(C, ray.kz), ray.shear);
}

// Would the reference's walk reach primitive `prim` of `inst` with the bound `t_max`?
// Its TLAS / BLAS boxes on the way are supersets of the primitive's own leaf box and the
// slab test is monotone, so it comes down to that leaf box (see commit.cpp, WalkTreeBuilder).
template <bool kAnalytic>
MCPT_HD bool reference_leaf_box_passes(const DeviceScene &sc, uint32_t inst, uint32_t prim, Ray ray, float t_max)
{
    ray.t_max = t_max;
    if (kAnalytic && sc.instances[inst].kind != kInstTriangles)
    {
        const size_t root = sc.instances[inst].blas_root; // a quadric is the single leaf of its BLAS
        return box_hit(sc.nodes[2 * root], sc.nodes[2 * root + 1], ray);
    }
    const float4 *p = sc.tri_pos + 3 * static_cast<size_t>(prim);
    const V3 lo = vmin(vmin(xyz(p[0]), xyz(p[1])), xyz(p[2])), hi = vmax(vmax(xyz(p[0]), xyz(p[1])), xyz(p[2]));
    return box_hit(float4{lo.x, lo.y, lo.z, 0.0f}, float4{hi.x, hi.y, hi.z, 0.0f}, ray); // triangle.cpp:9-15
}

// State of a closest query of the ordered walk.  ray.t_max is the CULLING bound of the
// box tests: slightly beyond the best distance, so that every primitive whose hit lies
// within rounding distance of the best one is still visited and compared.
struct ClosestState
{
    bool found;
    float best_t;
    uint32_t best_rank;
};

// One primitive of the ordered walk.  Shadow queries: accepted iff hit within the fixed
// bound.  Closest queries: the nearer hit wins; two hits within rounding distance of each
// other are decided by replaying what the reference does with that pair: it visits them
// in rank order, the first is accepted, the second only if its leaf box still passes with
// the first one's distance as the bound and its own distance is not larger
// (triangle.cpp:82) — and the later accepted one is kept.  This covers exact ties (shared
// edges, two surfaces in one plane) and the case where a flat box's entry distance and
// the triangle's own distance differ in the last bit.  Returns whether `hit` changed.
// kLeafCheck: the walk came here through a hierarchy whose boxes are LARGER than the exact ones (the 4-wide quantised
// form, device_scene.h: wide_nodes).  The exact hierarchy tests a primitive only when its own leaf box lets the ray in
// under the current bound; here that test is made explicitly, on the primitive's record, before the hit counts.  (A
// sliver's leaf box is a grown one in both hierarchies and its own rules below decide: not re-tested here.)
template <bool kAny, bool kAnalytic, bool kSlivers = true, bool kLeafCheck = false>
MCPT_HD bool test_slot(const DeviceScene &sc, uint32_t slot, Ray &ray, HitRaw &hit, ClosestState &best)
{
    const float4 *p = sc.walk_prims + 3 * static_cast<size_t>(slot);
    const uint32_t prim = as_uint(p[0].w), inst = as_uint(p[1].w), rank = as_uint(p[2].w); // rank | kWalkSliver
    SlotHit h;
    if (!kAnalytic || sc.instances[inst].kind == kInstTriangles)
    {
        h = triangle_probe(p, ray);
        if (kLeafCheck && h.hit && !(kSlivers && (rank & kWalkSliver)))
        {
            const V3 lo = vmin(vmin(xyz(p[0]), xyz(p[1])), xyz(p[2])), hi = vmax(vmax(xyz(p[0]), xyz(p[1])), xyz(p[2]));
            h.hit = box_hit(float4{lo.x, lo.y, lo.z, 0.0f}, float4{hi.x, hi.y, hi.z, 0.0f}, ray); // triangle.cpp:9-15, bound = ray.t_max
        }
    }
    else
    {
        const InstanceRec &rec = sc.instances[inst];
        Ray probe = ray;
        if (!kAny)
            probe.t_max = kMaxFloat;
        HitRaw cand;
        cand.a = cand.b = cand.c = 0.0f, cand.inside = false;
        uint32_t unused_rng = 0;
        if (rec.kind == kInstSphere)
            h.hit = sphere_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
        else if (rec.kind == kInstDisk)
            h.hit = disk_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
        else
            h.hit = cylinder_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
        h.t = probe.t_max, h.a = cand.a, h.b = cand.b, h.c = cand.c, h.inside = cand.inside;
        if (kLeafCheck && h.hit)
            h.hit = reference_leaf_box_passes<kAnalytic>(sc, inst, prim, ray, ray.t_max);
    }
    bool take;
    // A sliver's leaf box in the hierarchy is larger than the reference's (commit.cpp), so the walk
    // gets here in cases where the reference does not — e.g. when the ray STARTS on the sliver and its
    // flat box ends before t_min while the computed distance lands just above it, or passes along an
    // edge that neither neighbour's box admits.  Kernels for scenes without slivers are compiled
    // without all of it (kSlivers; it costs the VALU-bound cornell kernel 2 %).
    const bool slivers = kSlivers && sc.integrator.walk_sliver_reach > 0.0f;
    const bool sliver = slivers && (rank & kWalkSliver) != 0;
    if (kAny)
    {
        take = h.hit && !(h.t > ray.t_max);
        if (slivers && take && sliver) // reached through its grown box: would the reference's own leaf box let the ray in?
            take = reference_leaf_box_passes<kAnalytic>(sc, inst, prim, ray, ray.t_max);
    }
    else
    {
        // (a NaN distance — a ray with NaN components "hits" everything — is never kept: the
        //  reference drops such an instance hit when it merges it, tlas.cpp:27-33)
        take = h.hit && (best.found ? h.t < best.best_t : h.t <= ray.t_max);
        float tie = sc.integrator.walk_tie;
        bool reachable = true;
        if (slivers)
        {
            if ((rank | best.best_rank) & kWalkSliver)
                tie = sc.integrator.walk_sliver_reach; // what a sliver's distance can be off by
            if (h.hit && sliver)
                reachable = reference_leaf_box_passes<kAnalytic>(sc, inst, prim, ray, kMaxFloat);
        }
        if (!reachable)
            take = false; // the reference never tests it for this ray, whatever it has found so far
        else if (h.hit && best.found && fabsf(h.t - best.best_t) <= tie)
        {
            // rare: replay the reference on the pair (current best, this primitive) — hits within the
            // tie radius of each other
            if ((kSlivers ? (rank & ~kWalkSliver) > (best.best_rank & ~kWalkSliver) : rank > best.best_rank)) // the reference comes here second
                take = !(h.t > best.best_t) && reference_leaf_box_passes<kAnalytic>(sc, inst, prim, ray, best.best_t);
            else // the reference came here first; the current best is its second
                take = !(!(best.best_t > h.t) && reference_leaf_box_passes<kAnalytic>(sc, hit.inst, hit.prim, ray, h.t));
        }
    }
    if (take)
    {
        hit.inst = inst, hit.prim = prim, hit.a = h.a, hit.b = h.b, hit.c = h.c, hit.inside = h.inside;
        best.found = true;
        if (!kAny)
        {
            best.best_t = h.t, best.best_rank = rank;
            ray.t_max = h.t + ((kSlivers && (rank & kWalkSliver)) ? sc.integrator.walk_sliver_reach : sc.integrator.walk_tie);
        }
    }
    return take;
}

// The hit record of a closest query whose WINNER is already known (primary-visibility pre-pass): the winner's own
// test on the same ray gives the same barycentrics / object-space point, side and distance as it did inside the
// walk (test_slot copies exactly these values when it accepts a primitive).  Sets ray.t_max to the distance.
template <bool kAnalytic>
MCPT_HD void hit_from_record(const DeviceScene &sc, uint32_t inst, uint32_t prim, Ray &ray, HitRaw &hit)
{
    SlotHit h;
    if (!kAnalytic || sc.instances[inst].kind == kInstTriangles)
        h = triangle_probe(sc.tri_pos + 3 * static_cast<size_t>(prim), ray); // (walk_prims holds copies of these vertices)
    else
    {
        const InstanceRec &rec = sc.instances[inst];
        Ray probe = ray;
        probe.t_max = kMaxFloat;
        HitRaw cand;
        cand.a = cand.b = cand.c = 0.0f, cand.inside = false;
        uint32_t unused_rng = 0;
        if (rec.kind == kInstSphere)
            h.hit = sphere_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
        else if (rec.kind == kInstDisk)
            h.hit = disk_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
        else
            h.hit = cylinder_hit<false>(sc, sc.analytic[rec.analytic], prim, kNone, probe, unused_rng, cand);
        h.t = probe.t_max, h.a = cand.a, h.b = cand.b, h.c = cand.c, h.inside = cand.inside;
    }
    hit.inst = inst, hit.prim = prim, hit.a = h.a, hit.b = h.b, hit.c = h.c, hit.inside = h.inside;
    ray.t_max = h.t;
}

// True for exactly one of the currently active lanes of the wavefront.
MCPT_HD bool is_leading_lane()
{
#if MCPT_WAVE_DEVICE
    return static_cast<int>(__lane_id()) == __ffsll(static_cast<unsigned long long>(__ballot(1))) - 1;
#else
    return true;
#endif
}

#ifndef MCPT_FUSED_SLAB
#define MCPT_FUSED_SLAB 0 // (experiment switch: the slab test of the LDS-resident walk as one fma per plane, see walk_ordered)
#endif
#ifndef MCPT_SIGN_ADDRESSED_NODES
#define MCPT_SIGN_ADDRESSED_NODES 1 // measured: cornell 992 -> 1007 Msamples/s, frame unchanged (profiles/r02_experiments)
#endif

// Loop shape ("while-while"): an inner loop of pure node steps runs until the lane
// holds a primitive or has run out of work, then the primitive test runs.  On the
// GPU the wavefront therefore stays in the node phase until EVERY lane holds a
// primitive (or is done) before paying for one primitive phase; leaving the node
// phase earlier (when only a few lanes are still searching) was measured and is
// slower at every threshold (cornell: -2 % at 8 lanes ... -26 % at 64).
// (kStride: distance between a lane's consecutive stack entries — kWalkStackStride for the lane-interleaved stacks in LDS,
//  1 for a private array)
template <bool kAny, bool kAnalytic, bool kCount, bool kSlivers = true, uint32_t kStride = kWalkStackStride>
MCPT_HD bool walk_ordered(const DeviceScene &sc, uint32_t *stack, Ray &ray, HitRaw &hit, TraceStats &stats)
{
    if (sc.integrator.n_walk_nodes == 0)
        return false;
    ClosestState best{false, ray.t_max, 0};
    // entry 0 of the stack is a sentinel that reads as "no more work": popping never
    // has to test for an empty stack
    stack[0] = kWalkDone;
    uint32_t depth = 1; // entries on the stack
    uint32_t cur = 0;   // the top node
#if MCPT_SIGN_ADDRESSED_NODES && MCPT_WAVE_DEVICE
    const uint32_t near_x = ray.dir_rcp.x > 0 ? 0u : 4u, near_y = ray.dir_rcp.y > 0 ? 0u : 4u, near_z = ray.dir_rcp.z > 0 ? 0u : 4u;
#if MCPT_FUSED_SLAB
    // FUSED SLAB TEST (culling only).  The reference's plane distance is fl(fl(p - o) r): a subtraction and a product per
    // plane.  Here it is fma(p, r, c) with c = -(o r) -/+ s fixed per ray: one operation.  The two differ by roundings —
    // |fma(p, r, -fl(o r)) - fl(fl(p - o) r)| <= 3.01 eps |T| + 1.01 eps |o r| with T = (p - o) r the real value, |T| <= (M + |o|) |r|,
    // M = the largest |coordinate| of any box plane (IntegratorRec::walk_extent), eps = 2^-24 — so the slack
    // s = 2^-21 (M + 2 |o|) |r| (twice what the bound asks for, rounding of c and s included) makes every entry distance at most
    // and every exit distance at least the reference's: a box the reference's test lets the ray into is entered here too.  What
    // the looser test lets through in addition is stopped at the primitive by the exact leaf-box test (test_slot, kLeafCheck:
    // the reference reaches a primitive exactly when its own leaf box passes — interior boxes are supersets and the test is
    // monotone), so the answers are the reference's.
    const float slab = 0x1p-21f;
    const float qx = ray.origin.x * ray.dir_rcp.x, qy = ray.origin.y * ray.dir_rcp.y, qz = ray.origin.z * ray.dir_rcp.z;
    const float sx = slab * ((sc.integrator.walk_extent + 2.0f * fabsf(ray.origin.x)) * fabsf(ray.dir_rcp.x));
    const float sy = slab * ((sc.integrator.walk_extent + 2.0f * fabsf(ray.origin.y)) * fabsf(ray.dir_rcp.y));
    const float sz = slab * ((sc.integrator.walk_extent + 2.0f * fabsf(ray.origin.z)) * fabsf(ray.dir_rcp.z));
    const float cnx = -qx - sx, cfx = -qx + sx, cny = -qy - sy, cfy = -qy + sy, cnz = -qz - sz, cfz = -qz + sz;
#endif
#endif
    for (;;)
    {
        // ---- node steps until the lane holds a primitive or runs out of work ----
        while (!(cur & kWalkLeaf))
        {
            if (kCount)
            {
                stats.node_tests += 2;
                if (is_leading_lane())
                    ++stats.wave_node_steps;
            }
            float enter0, enter1;
            bool hit0, hit1;
            uint32_t ref0, ref1;
#if MCPT_SIGN_ADDRESSED_NODES && MCPT_WAVE_DEVICE
            {
                // (this walk runs on hierarchies staged in LDS.)  The slab test needs, per axis, the box plane the
                // ray enters through and the one it leaves through; which of lo / hi that is depends on the sign of
                // the ray's direction only.  Instead of loading the whole record and SELECTING (12 v_cndmask per
                // node step, a 4-cycle VALU class), the lane READS the right words: the record is 16 words
                // {lo0 ref0 | hi0 ref1 | lo1 - | hi1 -}, the entry plane of axis k is word k + near_k, near_k = 0 or 4.
                // Same operands, same arithmetic; the selects become LDS addresses.
                const float *w = reinterpret_cast<const float *>(sc.walk_nodes) + 16 * static_cast<size_t>(cur);
                const float *wnx = w + near_x, *wfx = w + (4u - near_x), *wny = w + near_y + 1, *wfy = w + (5u - near_y),
                            *wnz = w + near_z + 2, *wfz = w + (6u - near_z);
#if MCPT_FUSED_SLAB
                const float nx0 = __builtin_fmaf(wnx[0], ray.dir_rcp.x, cnx), fx0 = __builtin_fmaf(wfx[0], ray.dir_rcp.x, cfx);
                const float ny0 = __builtin_fmaf(wny[0], ray.dir_rcp.y, cny), fy0 = __builtin_fmaf(wfy[0], ray.dir_rcp.y, cfy);
                const float nz0 = __builtin_fmaf(wnz[0], ray.dir_rcp.z, cnz), fz0 = __builtin_fmaf(wfz[0], ray.dir_rcp.z, cfz);
                const float nx1 = __builtin_fmaf(wnx[8], ray.dir_rcp.x, cnx), fx1 = __builtin_fmaf(wfx[8], ray.dir_rcp.x, cfx);
                const float ny1 = __builtin_fmaf(wny[8], ray.dir_rcp.y, cny), fy1 = __builtin_fmaf(wfy[8], ray.dir_rcp.y, cfy);
                const float nz1 = __builtin_fmaf(wnz[8], ray.dir_rcp.z, cnz), fz1 = __builtin_fmaf(wfz[8], ray.dir_rcp.z, cfz);
#else
                const float nx0 = (wnx[0] - ray.origin.x) * ray.dir_rcp.x, fx0 = (wfx[0] - ray.origin.x) * ray.dir_rcp.x;
                const float ny0 = (wny[0] - ray.origin.y) * ray.dir_rcp.y, fy0 = (wfy[0] - ray.origin.y) * ray.dir_rcp.y;
                const float nz0 = (wnz[0] - ray.origin.z) * ray.dir_rcp.z, fz0 = (wfz[0] - ray.origin.z) * ray.dir_rcp.z;
                const float nx1 = (wnx[8] - ray.origin.x) * ray.dir_rcp.x, fx1 = (wfx[8] - ray.origin.x) * ray.dir_rcp.x;
                const float ny1 = (wny[8] - ray.origin.y) * ray.dir_rcp.y, fy1 = (wfy[8] - ray.origin.y) * ray.dir_rcp.y;
                const float nz1 = (wnz[8] - ray.origin.z) * ray.dir_rcp.z, fz1 = (wfz[8] - ray.origin.z) * ray.dir_rcp.z;
#endif
                enter0 = fmaxf(fmaxf(fmaxf(kEpsDistance, nx0), ny0), nz0);
                enter1 = fmaxf(fmaxf(fmaxf(kEpsDistance, nx1), ny1), nz1);
                hit0 = enter0 <= fminf(fminf(fminf(ray.t_max, fx0), fy0), fz0);
                hit1 = enter1 <= fminf(fminf(fminf(ray.t_max, fx1), fy1), fz1);
                ref0 = as_uint(w[3]), ref1 = as_uint(w[7]);
            }
#else
            {
                const float4 *n = sc.walk_nodes + 4 * static_cast<size_t>(cur);
                const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
                hit0 = box_enter(n0, n1, ray, enter0), hit1 = box_enter(n2, n3, ray, enter1);
                ref0 = as_uint(n0.w), ref1 = as_uint(n1.w);
            }
#endif
            // both hit: continue with the nearer child, postpone the other; one hit: go
            // there; none: take the most recently postponed reference.  Branch free: the
            // postponed reference is ALWAYS stored at the top (it only becomes part of the
            // stack when `both` advances depth) and the entry below the top is ALWAYS
            // loaded (depth >= 1 here: entry 0 is the sentinel) — two unconditional LDS
            // accesses cost less than the exec-mask regions of a three-way branch.
            const bool first0 = enter0 <= enter1, both = hit0 && hit1, none = !(hit0 || hit1);
            const uint32_t toward = (hit0 && (first0 || !hit1)) ? ref0 : ref1;
            const uint32_t postponed = stack[(depth - 1) * kStride];
            stack[depth * kStride] = first0 ? ref1 : ref0;
            depth = depth + (both ? 1u : 0u) - (none ? 1u : 0u);
            cur = none ? postponed : toward;
        }
        if (cur == kWalkDone)
            break;

        // ---- primitive test ---------------------------------------------------------
        if (kCount)
        {
            ++stats.prim_tests;
            if (is_leading_lane())
                ++stats.wave_prim_steps;
        }
#if MCPT_SIGN_ADDRESSED_NODES && MCPT_WAVE_DEVICE && MCPT_FUSED_SLAB
        constexpr bool kLeafCheck = true; // the boxes on the way here were tested loosely: the primitive's own box decides
#else
        constexpr bool kLeafCheck = false;
#endif
        if (test_slot<kAny, kAnalytic, kSlivers, kLeafCheck>(sc, cur & ~kWalkLeaf, ray, hit, best) && kAny)
            return true;
        --depth;
        cur = stack[depth * kStride];
    }
    if (!kAny)
        ray.t_max = best.found ? best.best_t : ray.t_max; // the exact distance, not the culling bound
    return best.found;
}

// Number of lanes of the wavefront for which `p` holds (1 or 0 on the host).
MCPT_HD uint32_t lanes_where(bool p)
{
#if MCPT_WAVE_DEVICE
    return static_cast<uint32_t>(__popcll(__ballot(p)));
#else
    return p ? 1u : 0u;
#endif
}

// Sum of `v` over the lanes of the wavefront (the value itself on the host).
MCPT_HD uint32_t lanes_sum(uint32_t v)
{
#if MCPT_WAVE_DEVICE
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
#endif
    return v;
}

// Wavefront aggregation helpers (a "wavefront" of the host build is one lane).
MCPT_HD uint32_t lane_rank_among(bool p, uint32_t &total) // index of this lane among the lanes where p holds
{
#if MCPT_WAVE_DEVICE
    const unsigned long long mask = __ballot(p);
    total = static_cast<uint32_t>(__popcll(mask));
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
#else
    total = p ? 1u : 0u;
    return 0u;
#endif
}

// One counter bump for all lanes of the wavefront where p holds: returns this lane's index in the
// reserved range [base, base + n).  (One LDS atomic per wavefront instead of one per lane.)
MCPT_HD uint32_t wave_reserve(uint32_t *counter, bool p)
{
    uint32_t n;
    const uint32_t rank = lane_rank_among(p, n);
#if MCPT_WAVE_DEVICE
    uint32_t base = 0;
    if (p && rank == 0)
        base = atomicAdd(counter, n);
    const unsigned long long mask = __ballot(p);
    if (mask == 0)
        return 0;
    base = __builtin_amdgcn_readlane(base, __ffsll(static_cast<long long>(mask)) - 1);
    return base + rank;
#else
    if (!p)
        return 0;
    const uint32_t base = *counter;
    *counter += n;
    return base + rank;
#endif
}

// The same walk for large scenes: a wavefront vote decides when to leave the node
// phase.  On a mesh a few lanes of a wavefront have rays that need many times the
// average number of node steps; waiting for them before every primitive phase
// (walk_ordered) leaves most lanes idle (node-phase lane utilisation 12 % on the
// matpreview scene).  Here the wavefront switches to the primitive phase as soon as
// fewer than sc.integrator.walk_break lanes are still searching (and somebody holds a
// primitive), or as soon as sc.integrator.walk_hold lanes hold one: the stragglers are
// parked with their cursor, the other lanes test their primitive, pop, and rejoin the
// node phase, so the stragglers' long walks overlap with the others' next steps
// (matpreview 266 -> 347 Msamples/s with walk_break 8..16, another 6 % with
// walk_hold 10..12).  With both 0 it is walk_ordered plus one ballot per step.  One
// lane's visiting order never changes.
template <bool kAny, bool kAnalytic, bool kCount, bool kSlivers = true>
MCPT_HD bool walk_ordered_vote(const DeviceScene &sc, uint32_t *stack, Ray &ray, HitRaw &hit, TraceStats &stats)
{
    if (sc.integrator.n_walk_nodes == 0)
        return false;
    ClosestState best{false, ray.t_max, 0};
    // entry 0 of the stack is a sentinel that reads as "no more work": popping never
    // has to test for an empty stack
    stack[0] = kWalkDone;
    uint32_t depth = 1; // entries on the stack
    uint32_t cur = 0;   // the top node
    const uint32_t break_below = sc.integrator.walk_break, hold_enough = sc.integrator.walk_hold;
    for (;;)
    {
        // ---- node phase: runs while enough lanes of the wavefront are searching ----
        for (;;)
        {
            const bool searching = !(cur & kWalkLeaf);
            const uint32_t n_searching = lanes_where(searching);
            if (n_searching == 0)
                break;
            if (break_below | hold_enough)
            {
                const uint32_t n_holding = lanes_where(cur != kWalkDone && !searching);
                if ((n_searching < break_below && n_holding != 0) || (hold_enough && n_holding >= hold_enough))
                    break;
            }
            if (!searching)
                continue;
            const float4 *n = sc.walk_nodes + 4 * static_cast<size_t>(cur);
            const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            if (kCount)
            {
                stats.node_tests += 2;
                if (is_leading_lane())
                    ++stats.wave_node_steps;
            }
            float enter0, enter1;
            const bool hit0 = box_enter(n0, n1, ray, enter0), hit1 = box_enter(n2, n3, ray, enter1);
            const uint32_t ref0 = as_uint(n0.w), ref1 = as_uint(n1.w);
            // both hit: continue with the nearer child, postpone the other; one hit: go
            // there; none: take the most recently postponed reference.  Branch free: the
            // postponed reference is ALWAYS stored at the top (it only becomes part of the
            // stack when `both` advances depth) and the entry below the top is ALWAYS
            // loaded (depth >= 1 here: entry 0 is the sentinel) — two unconditional LDS
            // accesses cost less than the exec-mask regions of a three-way branch.
            const bool first0 = enter0 <= enter1, both = hit0 && hit1, none = !(hit0 || hit1);
            const uint32_t toward = (hit0 && (first0 || !hit1)) ? ref0 : ref1;
            const uint32_t postponed = stack[(depth - 1) * kWalkStackStride];
            stack[depth * kWalkStackStride] = first0 ? ref1 : ref0;
            depth = depth + (both ? 1u : 0u) - (none ? 1u : 0u);
            cur = none ? postponed : toward;
        }
        if (lanes_where(cur != kWalkDone) == 0)
            break;
        if (cur == kWalkDone || !(cur & kWalkLeaf))
            continue; // parked: finished, or still searching while the others test primitives

        // ---- primitive test ---------------------------------------------------------
        if (kCount)
        {
            ++stats.prim_tests;
            if (is_leading_lane())
                ++stats.wave_prim_steps;
        }
        if (test_slot<kAny, kAnalytic, kSlivers>(sc, cur & ~kWalkLeaf, ray, hit, best) && kAny)
        {
            cur = kWalkDone;
            continue;
        }
        --depth;
        cur = stack[depth * kWalkStackStride];
    }
    if (!kAny)
        ray.t_max = best.found ? best.best_t : ray.t_max; // the exact distance, not the culling bound
    return best.found;
}

// Shading frame of the closest hit (second half of the reference's primitive
// tests: triangle.cpp:122-144, sphere.cpp:48-83, disk.cpp:46-108,
// cylinder.cpp:61-86), including bump mapping and the back-face flip.
// kPart: 0 = the whole record.  For instantiations WITHOUT textures (no bump map can turn the normal) the record can be built in two
// halves, with the same operations on the same operands: 1 = what path_resolve needs (instance, side, position, shading normal),
// 2 = the rest (uv, tangent, bitangent) of a record whose first half is in `s` — the class-sorted kernel builds it after its sort,
// for the lanes that stand on a surface, where the lanes of one BSDF kind share a wavefront (hip/sorted_kernel.hip).
template <bool kAnalytic, bool kTextures, int kPart>
MCPT_HD void make_surface_part(const DeviceScene &sc, const HitRaw &h, Surface &s)
{
    static_assert(kPart == 0 || !kTextures, "a bump map makes the normal depend on the tangent frame: one piece");
    constexpr bool kPoint = kPart != 2, kFrame = kPart != 1;
    if (kPoint)
        s.inst = h.inst, s.inside = h.inside;
    else if (h.inside)
        s.normal = -s.normal; // (back to the geometric side the frame is built on; flipped again below)
    const InstanceRec &rec = sc.instances[h.inst];
    const uint32_t bsdf = rec.bsdf;
    auto bump = [&](V3 &normal, V3 &tangent, V3 &bitangent)
    {
        if (bsdf == kNone)
            return;
        if (kTextures) // bsdf.cpp:238-253
        {
            const uint32_t map = sc.bsdfs[bsdf].bump;
            if (map != kNone)
            {
                const V2 g = texture_gradient(sc.textures, sc.texels, map, s.uv);
                normal = normalize(-g.u * tangent - g.v * bitangent + normal);
            }
        }
        bitangent = normalize(cross(normal, tangent));
        tangent = normalize(cross(bitangent, normal));
    };
    if (!kAnalytic || rec.kind == kInstTriangles)
    {
        const float4 *p = sc.tri_pos + 3 * static_cast<size_t>(h.prim);
        const float4 *at = sc.tri_attr + 9 * static_cast<size_t>(h.prim);
        const float u = h.a, v = h.b, w = h.c;
        if (kPoint)
        {
            s.position = u * xyz(p[0]) + v * xyz(p[1]) + w * xyz(p[2]);
            s.normal = normalize(u * xyz(at[0]) + v * xyz(at[1]) + w * xyz(at[2]));
        }
        if (kFrame)
        {
            s.uv = triangle_uv(at, u, v, w);
            s.tangent = normalize(u * xyz(at[3]) + v * xyz(at[4]) + w * xyz(at[5]));
            s.bitangent = normalize(u * xyz(at[6]) + v * xyz(at[7]) + w * xyz(at[8]));
            bump(s.normal, s.tangent, s.bitangent);
        }
        phase_mark(kPhaseTriangleFrame); // (diagnostic builds: phase_clock.h)
    }
    else
    {
        const AnalyticRec &q = sc.analytic[rec.analytic];
        const V3 p_local = V3{h.a, h.b, h.c};
        if (rec.kind == kInstSphere)
        {
            if (kPoint)
            {
                s.position = transform_point(q.to_world, p_local + from(q.center));
                s.normal = transform_dir(q.normal_to_world, normalize(p_local));
            }
            if (kFrame)
            {
                float theta, phi;
                s.uv = sphere_uv(p_local, theta, phi);
                constexpr float jitter = 0.01f * kPi;
                float theta_p = theta + jitter;
                const bool flip = theta_p > kPi;
                if (flip)
                    theta_p = theta - jitter;
                s.bitangent = normalize(transform_point(q.to_world, from_spherical(theta_p, phi, 1.0f)) - s.position);
                if (flip)
                    s.bitangent = -s.bitangent;
                s.tangent = normalize(cross(s.bitangent, s.normal));
                s.bitangent = normalize(cross(s.normal, s.tangent));
                bump(s.normal, s.tangent, s.bitangent);
            }
        }
        else if (rec.kind == kInstDisk)
        {
            V3 normal = V3{0, 0, 1};
            if (kPoint)
                s.position = transform_point(q.to_world, p_local);
            if (kFrame)
            {
                float theta, phi;
                to_spherical(p_local, theta, phi);
                const float r = length(p_local);
                s.uv = V2{r, phi * k1Div2Pi};
                constexpr float jitter = 0.01f * kPi;
                float r_p = r + jitter;
                const bool flip_b = r_p > r;
                if (flip_b)
                    r_p = r - jitter;
                float phi_p = phi + jitter;
                const bool flip_t = phi_p > kPi;
                if (flip_t)
                    phi_p = phi - jitter;
                const V3 e1 = from_spherical(theta, phi, r_p) - p_local, e2 = from_spherical(theta, phi_p, r) - p_local;
                const V2 duv1 = V2{r_p, s.uv.v} - s.uv, duv2 = V2{s.uv.u, phi_p * k1Div2Pi} - s.uv;
                const float norm = 1.0f / (duv2.u * duv1.v - duv1.u * duv2.v);
                V3 tangent = normalize((duv1.v * e2 - duv2.v * e1) * norm);
                if (flip_t)
                    tangent = -tangent;
                // the reference also derives a bitangent from the uv differentials
                // (disk.cpp:76-79) but overwrites it before use (disk.cpp:85)
                V3 bitangent = normalize(cross(normal, tangent));
                tangent = normalize(cross(bitangent, normal));
                bump(normal, tangent, bitangent);
                s.tangent = transform_dir(q.to_world, tangent);
                s.bitangent = transform_dir(q.to_world, bitangent);
            }
            if (kPoint) // (the local normal: (0, 0, 1) unless a bump map turned it — which only the one-piece form can have)
                s.normal = transform_dir(q.normal_to_world, normal);
        }
        else
        {
            if (kPoint)
            {
                s.position = transform_point(q.to_world, p_local);
                s.normal = transform_dir(q.normal_to_world, normalize(V3{p_local.x, p_local.y, 0.0f}));
            }
            if (kFrame)
            {
                s.uv = V2{gl::atan2f(p_local.y, p_local.x) * k1Div2Pi, p_local.z / q.length};
                s.tangent = transform_dir(q.normal_to_world, V3{0, 0, 1});
                s.bitangent = normalize(cross(s.normal, s.tangent));
                bump(s.normal, s.tangent, s.bitangent);
            }
        }
        phase_mark(kPhaseQuadricFrame);
    }
    if (h.inside)
    {
        s.normal = -s.normal;
        if (kFrame)
            s.bitangent = -s.bitangent;
    }
}

template <bool kAnalytic, bool kTextures>
MCPT_HD Surface make_surface(const DeviceScene &sc, const Ray &ray, const HitRaw &h)
{
    Surface s;
    make_surface_part<kAnalytic, kTextures, 0>(sc, h, s);
    return s;
}

// Uniform point on an instance for area-light sampling: area-weighted descent
// (blas.cpp:79-98) — right child of node i is skip(i + 1) — then the
// primitive's own sampler (triangle.cpp:150-160, sphere.cpp:88-105,
// disk.cpp:113-141, cylinder.cpp:91-104).
struct LightPoint
{
    V2 uv;
    V3 position, normal;
};

template <bool kAnalytic>
MCPT_HD LightPoint sample_instance(const DeviceScene &sc, uint32_t inst, float xi0, float xi1, float xi2)
{
    const InstanceRec &rec = sc.instances[inst];
    uint32_t node = rec.blas_root;
    float thresh = sc.node_area[node] * xi0;
    uint32_t object;
    while ((object = as_uint(sc.nodes[2 * static_cast<size_t>(node) + 1].w)) == kNoObject)
    {
        const uint32_t left = node + 1;
        const float left_area = sc.node_area[left];
        if (thresh < left_area)
        {
            node = left;
        }
        else
        {
            thresh -= left_area;
            node = as_uint(sc.nodes[2 * static_cast<size_t>(left)].w);
        }
    }
    LightPoint lp;
    if (!kAnalytic || rec.kind == kInstTriangles)
    {
        const float4 *p = sc.tri_pos + 3 * static_cast<size_t>(object);
        const float4 *at = sc.tri_attr + 9 * static_cast<size_t>(object);
        const float s = sqrtf(1.0f - xi1);
        const float u = 1.0f - s, v = s * xi2, w = 1.0f - u - v;
        lp.uv = triangle_uv(at, w, u, v);
        lp.position = w * xyz(p[0]) + u * xyz(p[1]) + v * xyz(p[2]);
        lp.normal = normalize(w * xyz(at[0]) + u * xyz(at[1]) + v * xyz(at[2]));
        return lp;
    }
    const AnalyticRec &q = sc.analytic[rec.analytic];
    if (rec.kind == kInstSphere)
    {
        const float cos_t = 1.0f - 2.0f * xi1;
        lp.uv = V2{xi2, gl::acosf(cos_t) * k1DivPi};
        const float sin_t = sqrtf(1.0f - sqr(cos_t)), phi = k2Pi * xi2;
        const V3 n_local = V3{sin_t * gl::cosf(phi), sin_t * gl::sinf(phi), cos_t};
        lp.position = transform_point(q.to_world, from(q.center) + q.radius * n_local);
        lp.normal = transform_dir(q.normal_to_world, n_local);
    }
    else if (rec.kind == kInstDisk)
    {
        const float r1 = 2.0f * xi1 - 1.0f, r2 = 2.0f * xi2 - 1.0f;
        float phi, r;
        if (r1 == 0.0f && r2 == 0.0f)
            r = phi = 0;
        else if (sqr(r1) > sqr(r2))
            r = r1, phi = kPiDiv4 * (r2 / r1);
        else
            r = r2, phi = kPiDiv2 - (r1 / r2) * kPiDiv4;
        lp.uv = V2{r, phi * k1Div2Pi};
        lp.position = transform_point(q.to_world, V3{(r * gl::cosf(phi)) * 0.5f, (r * gl::sinf(phi)) * 0.5f, 0});
        lp.normal = transform_dir(q.normal_to_world, V3{0, 0, 1});
    }
    else
    {
        const float phi = k2Pi * xi1, z = xi2 * q.length;
        lp.uv = V2{xi1, xi2};
        lp.position = transform_point(q.to_world, V3{gl::cosf(phi) * q.radius, gl::sinf(phi) * q.radius, z});
        lp.normal = transform_dir(q.normal_to_world, V3{gl::cosf(phi), gl::sinf(phi), 0});
    }
    return lp;
}

} // namespace mcpt

#endif // MCPT_TRAVERSAL_H
