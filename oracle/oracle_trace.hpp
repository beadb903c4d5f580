// TEST INFRASTRUCTURE — CPU oracle, part 4: rays, primitive tests, traversal.
//
// Restates reference src/rtcore: Ray set-up with the Woop shear constants
// (ray.cpp:18-47, WATERTIGHT_TRIANGLES build), the slab test (aabb.cpp:29-48),
// the watertight triangle test with its double-precision fallback
// (triangle.cpp:19-148), the analytic quadrics (sphere.cpp, disk.cpp,
// cylinder.cpp), the two-level stack traversal in the reference's visiting
// order — left child first, right child pushed, no near/far ordering
// (tlas.cpp:13-76, instance.cpp:25-54, blas.cpp:18-77) — and the
// area-weighted point sampling of an instance (blas.cpp:79-98).
#ifndef ORACLE_TRACE_HPP
#define ORACLE_TRACE_HPP

#include "oracle_shading.hpp"

namespace orc
{

struct Ray // ray.hpp:9-27
{
    float t_min = kEpsDistance, t_max = kMaxF;
    int k[3] = {2, 0, 1};
    V3 shear, origin, dir, dir_rcp;
};

inline Ray MakeRay(V3 origin, V3 dir) // ray.cpp:18-47
{
    Ray r;
    r.origin = origin, r.dir = dir;
    for (int i = 0; i < 3; ++i)
        r.dir_rcp[i] = 1.0f / (dir[i] != 0 ? dir[i] : kEpsDistance);
    const float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    r.k[2] = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
    r.k[0] = r.k[2] + 1;
    if (r.k[0] == 3)
        r.k[0] = 0;
    r.k[1] = r.k[0] + 1;
    if (r.k[1] == 3)
        r.k[1] = 0;
    if (dir[r.k[2]] < 0.0f)
    {
        const int t = r.k[0];
        r.k[0] = r.k[1];
        r.k[1] = t;
    }
    r.shear = {dir[r.k[0]] / dir[r.k[2]], dir[r.k[1]] / dir[r.k[2]], 1.0f / dir[r.k[2]]};
    return r;
}

struct Hit // hit.hpp:9-30
{
    bool valid = false, inside = false;
    uint32_t inst = kNone, prim = kNone, med_int = kNone, med_ext = kNone;
    V2 uv;
    V3 position, normal, tangent, bitangent;
};

inline bool BoxHit(const Box &b, const Ray &r) // aabb.cpp:29-48
{
    const V3 t0 = (b.lo - r.origin) * r.dir_rcp, t1 = (b.hi - r.origin) * r.dir_rcp;
    float t_enter = r.t_min, t_exit = r.t_max;
    for (int i = 0; i < 3; ++i)
    {
        if (r.dir_rcp[i] > 0)
        {
            t_enter = fmaxf(t_enter, t0[i]);
            t_exit = fminf(t_exit, t1[i]);
        }
        else
        {
            t_enter = fmaxf(t_enter, t1[i]);
            t_exit = fminf(t_exit, t0[i]);
        }
    }
    return t_enter <= t_exit;
}

// Shared tail of every primitive test: opacity, shrink t_max, build the
// shading frame with bump mapping and back-face flip.
inline void FinishFrame(const Scene &sc, const Bsdf *bsdf, V2 uv, bool inside,
                        V3 *normal, V3 *tangent, V3 *bitangent)
{
    if (bsdf != nullptr)
    {
        *normal = BsdfBump(sc, *bsdf, *normal, *tangent, *bitangent, uv);
        *bitangent = Unit(Cross(*normal, *tangent));
        *tangent = Unit(Cross(*bitangent, *normal));
    }
    if (inside)
    {
        *normal = -*normal;
        *bitangent = -*bitangent;
    }
}

inline bool TriangleHit(const Scene &sc, const Prim &q, const Bsdf *bsdf,
                        uint32_t *rng, Ray *ray, Hit *hit) // triangle.cpp:19-148 (Woop branch)
{
    const V3 A = q.p[0] - ray->origin, B = q.p[1] - ray->origin, C = q.p[2] - ray->origin;
    const int kx = ray->k[0], ky = ray->k[1], kz = ray->k[2];
    const float Ax = A[kx] - ray->shear.x * A[kz], Ay = A[ky] - ray->shear.y * A[kz];
    const float Bx = B[kx] - ray->shear.x * B[kz], By = B[ky] - ray->shear.y * B[kz];
    const float Cx = C[kx] - ray->shear.x * C[kz], Cy = C[ky] - ray->shear.y * C[kz];
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f)
    {
        U = static_cast<float>(static_cast<double>(Cx) * static_cast<double>(By) -
                               static_cast<double>(Cy) * static_cast<double>(Bx));
        V = static_cast<float>(static_cast<double>(Ax) * static_cast<double>(Cy) -
                               static_cast<double>(Ay) * static_cast<double>(Cx));
        W = static_cast<float>(static_cast<double>(Bx) * static_cast<double>(Ay) -
                               static_cast<double>(By) * static_cast<double>(Ax));
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    const float det = U + V + W;
    if (det == 0.0f)
        return false;
    const float Az = ray->shear.z * A[kz], Bz = ray->shear.z * B[kz], Cz = ray->shear.z * C[kz];
    const float T = U * Az + V * Bz + W * Cz;
    const float det_inv = 1.0f / det;
    const float t = T * det_inv;
    if (t > ray->t_max || t < ray->t_min)
        return false;
    const float u = U * det_inv, v = V * det_inv, w = W * det_inv;

    const V2 uv = Bary(q.uv, u, v, w);
    if (bsdf != nullptr && BsdfTransparent(sc, *bsdf, uv, rng))
        return false;
    ray->t_max = t;
    if (hit != nullptr)
    {
        const bool inside = det_inv < 0;
        const V3 position = Bary(q.p, u, v, w);
        V3 normal = Unit(Bary(q.n, u, v, w)), tangent = Unit(Bary(q.t, u, v, w)),
           bitangent = Unit(Bary(q.b, u, v, w));
        FinishFrame(sc, bsdf, uv, inside, &normal, &tangent, &bitangent);
        *hit = Hit();
        hit->valid = true, hit->inside = inside, hit->prim = q.local_id;
        hit->uv = uv, hit->position = position, hit->normal = normal;
        hit->tangent = tangent, hit->bitangent = bitangent;
    }
    return true;
}

inline bool SphereHit(const Scene &sc, const Prim &q, const Bsdf *bsdf,
                      uint32_t *rng, Ray *ray, Hit *hit) // sphere.cpp:17-86
{
    const M4 to_local = Inverse(q.to_world);
    const V3 o = XformPoint(to_local, ray->origin) - q.center,
             d = XformDir(to_local, ray->dir);
    const float a = Dot(d, d), b = 2.0f * Dot(d, o), c = Dot(o, o) - Sq(q.radius);
    float t_near = 0.0f, t_far = 0.0f;
    if (!Quadratic(a, b, c, &t_near, &t_far) || t_far < kEpsDistance)
        return false;
    float t = t_near < kEpsDistance ? t_far : t_near;
    const V3 p_local = o + t * d, position = XformPoint(q.to_world, p_local + q.center);
    t = Len(position - ray->origin);
    if (t > ray->t_max || t < ray->t_min)
        return false;
    float theta, phi;
    ToSpherical(p_local, &theta, &phi, nullptr);
    const V2 uv = {phi * k1Div2Pi, theta * k1DivPi};
    if (bsdf != nullptr && BsdfTransparent(sc, *bsdf, uv, rng))
        return false;
    ray->t_max = t;
    if (hit != nullptr)
    {
        const bool inside = c < 0.0f;
        const M4 normal_to_world = Inverse(Transpose(q.to_world));
        const V3 n_local = Unit(p_local);
        V3 normal = XformDir(normal_to_world, n_local);
        constexpr float jitter = 0.01f * kPi;
        float theta_p = theta + jitter;
        const bool flip = theta_p > kPi;
        if (flip)
            theta_p = theta - jitter;
        const V3 p_prime = XformPoint(q.to_world, FromSpherical(theta_p, phi, 1));
        V3 bitangent = Unit(p_prime - position);
        if (flip)
            bitangent = -bitangent;
        V3 tangent = Unit(Cross(bitangent, normal));
        bitangent = Unit(Cross(normal, tangent));
        FinishFrame(sc, bsdf, uv, inside, &normal, &tangent, &bitangent);
        *hit = Hit();
        hit->valid = true, hit->inside = inside, hit->prim = q.local_id;
        hit->uv = uv, hit->position = position, hit->normal = normal;
        hit->tangent = tangent, hit->bitangent = bitangent;
    }
    return true;
}

inline bool DiskHit(const Scene &sc, const Prim &q, const Bsdf *bsdf,
                    uint32_t *rng, Ray *ray, Hit *hit) // disk.cpp:17-111
{
    const M4 to_local = Inverse(q.to_world);
    const V3 o = XformPoint(to_local, ray->origin), d = XformDir(to_local, ray->dir);
    const float t_z = -o.z / d.z;
    if (t_z < kEpsFloat)
        return false;
    const V3 p_local = o + t_z * d;
    if (Len(p_local) > 0.5f)
        return false;
    const V3 position = XformPoint(q.to_world, p_local);
    const float t = Len(position - ray->origin);
    if (t > ray->t_max || t < ray->t_min)
        return false;
    float theta, phi, r;
    ToSpherical(p_local, &theta, &phi, &r);
    const V2 uv = {r, phi * k1Div2Pi};
    if (bsdf != nullptr && BsdfTransparent(sc, *bsdf, uv, rng))
        return false;
    ray->t_max = t;
    if (hit != nullptr)
    {
        const bool inside = d.z > 0;
        constexpr float jitter = 0.01f * kPi;
        float r_p = r + jitter;
        const bool flip_b = r_p > r;
        if (flip_b)
            r_p = r - jitter;
        float phi_p = phi + jitter;
        const bool flip_t = phi_p > kPi;
        if (flip_t)
            phi_p = phi - jitter;
        const V3 e1 = FromSpherical(theta, phi, r_p) - p_local,
                 e2 = FromSpherical(theta, phi_p, r) - p_local;
        const V2 duv1 = V2{r_p, uv.v} - uv, duv2 = V2{uv.u, phi_p * k1Div2Pi} - uv;
        const float norm = 1.0f / (duv2.u * duv1.v - duv1.u * duv2.v);
        V3 tangent = Unit((duv1.v * e2 - duv2.v * e1) * norm),
           bitangent = Unit((duv2.u * e1 - duv1.u * e2) * norm), normal = {0, 0, 1};
        if (flip_b)
            bitangent = -bitangent;
        if (flip_t)
            tangent = -tangent;
        bitangent = Unit(Cross(normal, tangent));
        tangent = Unit(Cross(bitangent, normal));
        if (bsdf != nullptr)
        {
            normal = BsdfBump(sc, *bsdf, normal, tangent, bitangent, uv);
            bitangent = Unit(Cross(normal, tangent));
            tangent = Unit(Cross(bitangent, normal));
        }
        const M4 normal_to_world = Inverse(Transpose(q.to_world));
        normal = XformDir(normal_to_world, normal);
        tangent = XformDir(q.to_world, tangent);
        bitangent = XformDir(q.to_world, bitangent);
        if (inside)
        {
            normal = -normal;
            bitangent = -bitangent;
        }
        *hit = Hit();
        hit->valid = true, hit->inside = inside, hit->prim = q.local_id;
        hit->uv = uv, hit->position = position, hit->normal = normal;
        hit->tangent = tangent, hit->bitangent = bitangent;
    }
    return true;
}

inline bool CylinderHit(const Scene &sc, const Prim &q, const Bsdf *bsdf,
                        uint32_t *rng, Ray *ray, Hit *hit) // cylinder.cpp:21-89
{
    const M4 to_local = Inverse(q.to_world);
    const V3 o = XformPoint(to_local, ray->origin), d = XformDir(to_local, ray->dir);
    const float a = Sq(d.x) + Sq(d.y), b = 2.0f * (d.x * o.x + d.y * o.y),
                c = Sq(o.x) + Sq(o.y) - Sq(q.radius);
    float t_near = 0.0f, t_far = 0.0f;
    if (!Quadratic(a, b, c, &t_near, &t_far) || t_far < kEpsDistance)
        return false;
    const float z_near = o.z + d.z * t_near, z_far = o.z + d.z * t_far;
    float t = 0;
    if (kEpsDistance < t_near && 0.0f <= z_near && z_near <= q.length)
        t = t_near;
    else if (0.0 <= z_far && z_far <= q.length)
        t = t_far;
    else
        return false;
    const V3 p_local = o + t * d;
    const V2 uv = {atan2f(p_local.y, p_local.x) * k1Div2Pi, p_local.z / q.length};
    if (bsdf != nullptr && BsdfTransparent(sc, *bsdf, uv, rng))
        return false;
    const V3 position = XformPoint(q.to_world, p_local);
    t = Len(position - ray->origin);
    if (t > ray->t_max || t < ray->t_min)
        return false;
    ray->t_max = t;
    if (hit != nullptr)
    {
        const bool inside = c < 0.0f;
        const M4 normal_to_world = Inverse(Transpose(q.to_world));
        const V3 n_local = Unit(V3{p_local.x, p_local.y, 0.0f});
        V3 normal = XformDir(normal_to_world, n_local),
           tangent = XformDir(normal_to_world, {0, 0, 1}),
           bitangent = Unit(Cross(normal, tangent));
        FinishFrame(sc, bsdf, uv, inside, &normal, &tangent, &bitangent);
        *hit = Hit();
        hit->valid = true, hit->inside = inside, hit->prim = q.local_id;
        hit->uv = uv, hit->position = position, hit->normal = normal;
        hit->tangent = tangent, hit->bitangent = bitangent;
    }
    return true;
}

inline bool PrimHit(const Scene &sc, const Prim &q, const Bsdf *bsdf,
                    uint32_t *rng, Ray *ray, Hit *hit) // primitive.cpp:84-103
{
    switch (q.shape)
    {
    case Shape::kTriangle: return TriangleHit(sc, q, bsdf, rng, ray, hit);
    case Shape::kSphere: return SphereHit(sc, q, bsdf, rng, ray, hit);
    case Shape::kDisk: return DiskHit(sc, q, bsdf, rng, ray, hit);
    case Shape::kCylinder: return CylinderHit(sc, q, bsdf, rng, ray, hit);
    }
    return false;
}

// Counters for the measurement rows of SURVEY.md §8(d) (optional).
struct TraceStats
{
    uint64_t closest_rays = 0, shadow_rays = 0, node_tests = 0, prim_tests = 0;
};

inline const Bsdf *InstanceBsdf(const Scene &sc, uint32_t inst)
{
    const uint32_t id = sc.inst_bsdf[inst];
    return id == kNone ? nullptr : &sc.bsdfs[id];
}

// blas.cpp:18-44 / 46-77: returns true as soon as `any` is set and a
// primitive accepts.
inline bool WalkBlas(const Scene &sc, const Inst &it, const Bsdf *bsdf,
                     uint32_t *rng, Ray *ray, Hit *hit, bool any, TraceStats *st)
{
    const Node *nodes = sc.nodes.data() + it.node_base;
    const Prim *prims = sc.prims.data() + it.prim_base;
    uint32_t stack[65];
    stack[0] = 0;
    int top = 0;
    while (top >= 0)
    {
        const Node *n = nodes + stack[top--];
        for (;;)
        {
            if (st)
                ++st->node_tests;
            if (!BoxHit(n->box, *ray))
                break;
            if (n->leaf)
            {
                if (st)
                    ++st->prim_tests;
                const bool accepted = PrimHit(sc, prims[n->object], bsdf, rng, ray, any ? nullptr : hit);
                if (any && accepted)
                    return true;
                break;
            }
            stack[++top] = n->right;
            n = nodes + n->left;
        }
    }
    return false;
}

// tlas.cpp:13-42 + instance.cpp:25-44.
inline Hit ClosestHit(const Scene &sc, uint32_t *rng, Ray *ray, TraceStats *st = nullptr)
{
    Hit best;
    if (st)
        ++st->closest_rays;
    if (sc.insts.empty())
        return best;
    const Node *nodes = sc.nodes.data();
    uint32_t stack[65];
    stack[0] = 0;
    int top = 0;
    while (top >= 0)
    {
        const Node *n = nodes + stack[top--];
        for (;;)
        {
            if (st)
                ++st->node_tests;
            if (!BoxHit(n->box, *ray))
                break;
            if (n->leaf)
            {
                const uint32_t id = n->object;
                const Inst &it = sc.insts[id];
                Ray local = *ray;
                Hit h;
                WalkBlas(sc, it, InstanceBsdf(sc, id), rng, &local, &h, false, st);
                if (h.valid && local.t_max <= ray->t_max)
                {
                    *ray = local;
                    best = h;
                    best.inst = id;
                    best.med_int = it.med_int;
                    best.med_ext = it.med_ext;
                }
                break;
            }
            stack[++top] = n->right;
            n = nodes + n->left;
        }
    }
    return best;
}

// tlas.cpp:43-75 + instance.cpp:46-54.
inline bool AnyHit(const Scene &sc, uint32_t *rng, Ray *ray, TraceStats *st = nullptr)
{
    if (st)
        ++st->shadow_rays;
    if (sc.insts.empty())
        return false;
    const Node *nodes = sc.nodes.data();
    uint32_t stack[65];
    stack[0] = 0;
    int top = 0;
    while (top >= 0)
    {
        const Node *n = nodes + stack[top--];
        for (;;)
        {
            if (st)
                ++st->node_tests;
            if (!BoxHit(n->box, *ray))
                break;
            if (n->leaf)
            {
                const uint32_t id = n->object;
                if (WalkBlas(sc, sc.insts[id], InstanceBsdf(sc, id), rng, ray, nullptr, true, st))
                    return true;
                break;
            }
            stack[++top] = n->right;
            n = nodes + n->left;
        }
    }
    return false;
}

// Uniform point on an instance: area-weighted descent of its BLAS
// (blas.cpp:79-98), then the primitive's own sampler.
inline Hit SampleInstance(const Scene &sc, uint32_t inst, float xi0, float xi1, float xi2)
{
    const Inst &it = sc.insts[inst];
    const Node *nodes = sc.nodes.data() + it.node_base;
    const Node *n = nodes;
    float thresh = n->area * xi0;
    while (!n->leaf)
    {
        if (thresh < nodes[n->left].area)
        {
            n = nodes + n->left;
        }
        else
        {
            thresh -= nodes[n->left].area;
            n = nodes + n->right;
        }
    }
    const Prim &q = sc.prims[it.prim_base + n->object];
    Hit h;
    h.valid = true, h.prim = q.local_id;
    switch (q.shape)
    {
    case Shape::kTriangle: // triangle.cpp:150-160
    {
        const float s = sqrtf(1.0f - xi1);
        const float u = 1.0f - s, v = s * xi2, w = 1.0f - u - v;
        h.uv = Bary(q.uv, w, u, v);
        h.position = Bary(q.p, w, u, v);
        h.normal = Unit(Bary(q.n, w, u, v));
        break;
    }
    case Shape::kSphere: // sphere.cpp:88-105
    {
        const float cos_t = 1.0f - 2.0f * xi1;
        h.uv = {xi2, acosf(cos_t) * k1DivPi};
        const float sin_t = sqrtf(1.0f - Sq(cos_t)), phi = k2Pi * xi2;
        const V3 n_local = {sin_t * cosf(phi), sin_t * sinf(phi), cos_t},
                 p_local = q.center + q.radius * n_local;
        h.position = XformPoint(q.to_world, p_local);
        h.normal = XformDir(Inverse(Transpose(q.to_world)), n_local);
        break;
    }
    case Shape::kDisk: // disk.cpp:113-141
    {
        const float r1 = 2.0f * xi1 - 1.0f, r2 = 2.0f * xi2 - 1.0f;
        float phi, r;
        if (r1 == 0.0f && r2 == 0.0f)
        {
            r = phi = 0;
        }
        else if (Sq(r1) > Sq(r2))
        {
            r = r1;
            phi = kPiDiv4 * (r2 / r1);
        }
        else
        {
            r = r2;
            phi = kPiDiv2 - (r1 / r2) * kPiDiv4;
        }
        const float px = r * cosf(phi), py = r * sinf(phi);
        h.uv = {r, phi * k1Div2Pi};
        h.position = XformPoint(q.to_world, {px * 0.5f, py * 0.5f, 0});
        h.normal = XformDir(Inverse(Transpose(q.to_world)), {0, 0, 1});
        break;
    }
    case Shape::kCylinder: // cylinder.cpp:91-104
    {
        const float phi = k2Pi * xi1, z = xi2 * q.length;
        h.uv = {xi1, xi2};
        h.position = XformPoint(q.to_world, {cosf(phi) * q.radius, sinf(phi) * q.radius, z});
        h.normal = XformDir(Inverse(Transpose(q.to_world)), {cosf(phi), sinf(phi), 0});
        break;
    }
    }
    return h;
}

} // namespace orc

#endif // ORACLE_TRACE_HPP
