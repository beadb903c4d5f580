// Vertex normals and tangent frames for meshes read from OBJ / PLY files, the way the
// reference gets them from its importer.
//
// The reference imports these formats with assimp (an external dependency, found by
// find_package, no version pinned; src/parser/model_loader.cpp:506-529) and asks for
// aiProcess_GenSmoothNormals (unless face_normals) and aiProcess_CalcTangentSpace.  Its
// renderer then takes the importer's per-vertex tangents whenever they exist
// (src/rtcore/scene.cpp:93-100) — which matters: the scenes it ships (classroom,
// dining-room, ...) write "vt 0 0" for every corner of most meshes, and the renderer's own
// UV-derived tangent (scene.cpp:63-80) divides by the zero UV determinant there: NaN frames
// on every such triangle.  assimp is not available here, so this file restates the two
// post-processing steps from their published algorithm (assimp 5.x:
// GenVertexNormalsProcess with the default 175 degree limit = no angle limit,
// CalcTangentsProcess with the default 45 degree smoothing limit, both on top of the
// SpatialSort position finder with epsilon = 1e-4 * bounding-box diagonal).  PARITY
// UNPINNED: there is no assimp build to compare with; frames built from these tangents
// are valid and orthonormal, a render is statistically the reference's, not pixel-exact
// (the tangent's direction steers every sampled direction).  DESIGN.md §8.
#include "asset_io.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace mcpt
{
namespace
{

struct F3
{
    float x = 0, y = 0, z = 0;
};
inline F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float Dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F3 Cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float Length(F3 a) { return std::sqrt(Dot(a, a)); }
inline F3 Normalized(F3 a) { return a * (1.0f / Length(a)); } // 0 -> NaN, like aiVector3D::Normalize
inline F3 NormalizedSafe(F3 a)
{
    const float len = Length(a);
    return len > 0.0f ? a * (1.0f / len) : a;
}
inline bool Special(float v) { return std::isnan(v) || std::isinf(v); }
inline bool Special(F3 a) { return Special(a.x) || Special(a.y) || Special(a.z); }
inline F3 At(const std::vector<float> &v, size_t i) { return {v[3 * i], v[3 * i + 1], v[3 * i + 2]}; }
inline void Put(std::vector<float> &v, size_t i, F3 a) { v[3 * i] = a.x, v[3 * i + 1] = a.y, v[3 * i + 2] = a.z; }

// All vertices within a radius of a position: vertices sorted by their distance to a
// fixed oblique plane; a query scans the slab [d - r, d + r) and keeps those closer than r.
class PositionFinder
{
  public:
    explicit PositionFinder(const std::vector<float> &positions) : positions_(positions)
    {
        const size_t n = positions.size() / 3;
        normal_ = Normalized({0.8523f, 0.34321f, 0.5736f});
        entries_.resize(n);
        F3 lo{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
        F3 hi{-lo.x, -lo.y, -lo.z};
        for (size_t i = 0; i < n; ++i)
        {
            const F3 p = At(positions, i);
            entries_[i] = {Dot(p, normal_), static_cast<uint32_t>(i)};
            lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
            hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
        }
        std::stable_sort(entries_.begin(), entries_.end(), [](const Entry &a, const Entry &b) { return a.d < b.d; });
        epsilon_ = n ? Length(hi - lo) * 1e-4f : 0.0f;
    }
    float epsilon() const { return epsilon_; }
    void Find(F3 p, float radius, std::vector<uint32_t> &out) const
    {
        out.clear();
        const float d = Dot(p, normal_), lo = d - radius, hi = d + radius, r2 = radius * radius;
        auto it = std::lower_bound(entries_.begin(), entries_.end(), lo, [](const Entry &e, float v) { return e.d < v; });
        for (; it != entries_.end() && it->d < hi; ++it)
        {
            const F3 q = At(positions_, it->index) - p;
            if (Dot(q, q) < r2)
                out.push_back(it->index);
        }
    }

  private:
    struct Entry
    {
        float d;
        uint32_t index;
    };
    const std::vector<float> &positions_;
    std::vector<Entry> entries_;
    F3 normal_;
    float epsilon_ = 0;
};

} // namespace

// GenVertexNormalsProcess without an angle limit: every vertex gets the face normal of
// (the last of) its faces, then all vertices at one position get the normalised sum.
void GenerateSmoothNormals(MeshData &m)
{
    const size_t n = m.positions.size() / 3;
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    std::vector<float> face(3 * n, qnan);
    for (size_t t = 0; t + 2 < m.indices.size(); t += 3)
    {
        const F3 a = At(m.positions, m.indices[t]), b = At(m.positions, m.indices[t + 1]), c = At(m.positions, m.indices[t + 2]);
        const F3 normal = NormalizedSafe(Cross(b - a, c - a));
        for (int k = 0; k < 3; ++k)
            Put(face, m.indices[t + k], normal);
    }
    const PositionFinder finder(m.positions);
    m.normals.assign(3 * n, 0.0f);
    std::vector<bool> done(n, false);
    std::vector<uint32_t> found;
    for (size_t i = 0; i < n; ++i)
    {
        if (done[i])
            continue;
        finder.Find(At(m.positions, i), finder.epsilon(), found);
        F3 sum;
        for (const uint32_t v : found)
            if (!std::isnan(face[3 * v]))
                sum = sum + At(face, v);
        sum = NormalizedSafe(sum);
        for (const uint32_t v : found)
            Put(m.normals, v, sum), done[v] = true;
        if (found.empty()) // a mesh without extent: epsilon 0 finds nothing
            Put(m.normals, i, NormalizedSafe(Special(At(face, i)) ? F3{} : At(face, i)));
    }
}

// CalcTangentsProcess: per face, the directions of +u and +v in model space (or, where the
// three texture coordinates do not span an area, the triangle's two edges), projected into
// each corner's normal plane and orthogonalised; then vertices at one position with
// (nearly) the same normal and tangent directions within 45 degrees share the average.
void CalcTangentSpace(MeshData &m)
{
    const size_t n = m.positions.size() / 3;
    m.tangents.clear(), m.bitangents.clear();
    if (m.normals.size() != 3 * n || m.texcoords.size() != 2 * n || n == 0)
        return; // assimp: "failed to compute tangents; need UV data / normals" — none are produced
    m.tangents.assign(3 * n, 0.0f), m.bitangents.assign(3 * n, 0.0f);
    for (size_t t = 0; t + 2 < m.indices.size(); t += 3)
    {
        const uint32_t p0 = m.indices[t], p1 = m.indices[t + 1], p2 = m.indices[t + 2];
        const F3 v = At(m.positions, p1) - At(m.positions, p0), w = At(m.positions, p2) - At(m.positions, p0);
        float sx = m.texcoords[2 * p1] - m.texcoords[2 * p0], sy = m.texcoords[2 * p1 + 1] - m.texcoords[2 * p0 + 1];
        float tx = m.texcoords[2 * p2] - m.texcoords[2 * p0], ty = m.texcoords[2 * p2 + 1] - m.texcoords[2 * p0 + 1];
        const float flip = (tx * sy - ty * sx) < 0.0f ? -1.0f : 1.0f;
        if (sx * ty == sy * tx)
            sx = 0.0f, sy = 1.0f, tx = 1.0f, ty = 0.0f;
        const F3 tangent = (w * sy - v * ty) * flip, bitangent = (v * tx - w * sx) * flip;
        for (const uint32_t p : {p0, p1, p2})
        {
            const F3 normal = At(m.normals, p);
            F3 lt = tangent - normal * Dot(tangent, normal);
            F3 lb = bitangent - normal * Dot(bitangent, normal) - lt * Dot(bitangent, lt);
            lt = NormalizedSafe(lt), lb = NormalizedSafe(lb);
            const bool bad_t = Special(lt), bad_b = Special(lb);
            if (bad_t != bad_b)
            {
                if (bad_t)
                    lt = NormalizedSafe(Cross(normal, lb));
                else
                    lb = NormalizedSafe(Cross(lt, normal));
            }
            Put(m.tangents, p, lt), Put(m.bitangents, p, lb);
        }
    }
    const PositionFinder finder(m.positions);
    const float same_normal = 0.9999f, limit = std::cos(45.0f * 3.14159265358979323846f / 180.0f);
    std::vector<bool> done(n, false);
    std::vector<uint32_t> found, group;
    for (size_t a = 0; a < n; ++a)
    {
        if (done[a])
            continue;
        const F3 normal = At(m.normals, a), tangent = At(m.tangents, a), bitangent = At(m.bitangents, a);
        finder.Find(At(m.positions, a), finder.epsilon(), found);
        group.clear();
        group.push_back(static_cast<uint32_t>(a));
        for (const uint32_t idx : found) // (finds `a` again: it enters its own average twice, as in assimp)
        {
            if (done[idx] || Dot(At(m.normals, idx), normal) < same_normal || Dot(At(m.tangents, idx), tangent) < limit ||
                Dot(At(m.bitangents, idx), bitangent) < limit)
                continue;
            group.push_back(idx);
            done[idx] = true;
        }
        F3 st, sb;
        for (const uint32_t idx : group)
            st = st + At(m.tangents, idx), sb = sb + At(m.bitangents, idx);
        st = Normalized(st), sb = Normalized(sb);
        for (const uint32_t idx : group)
            Put(m.tangents, idx, st), Put(m.bitangents, idx, sb);
    }
}

} // namespace mcpt
