// Stand-in geometry for mesh files that a scene file names but that are not on disk.
//
// Why: the reference repository ships resources/scene/dragon/scene.xml (BASELINE config 3) without four
// of its sixteen OBJ files (/root/reference/.MISSING_LARGE_BLOBS: Mesh007, Mesh008, Mesh012, Mesh013 —
// the dragon's body and two ground pieces, ~750 k of the scene's ~800 k triangles).  The reference's
// loader fails on a missing file (model_loader.cpp:440-447) and so does LoadXmlScene by default.  With a
// stand-in table the scene file, its camera, materials, emitter and its twelve real meshes are used as
// they are and each missing file is replaced by a deterministic procedural mesh of a stated size, so
// that the configuration can be rendered and measured with its real triangle count.  A stand-in is NOT
// the missing asset: results on such a scene say "stand-in" wherever they are quoted.
//
// Table format, one line per file (blank lines and lines starting with '#' are skipped):
//     <file name as written in the XML>  blob   cx cy cz  rx ry rz  rings segments  seed amplitude
//     <file name as written in the XML>  sheet  x0 z0 x1 z1  y  nx nz  seed amplitude
//     <file name as written in the XML>  disc   cx cz  r0 r1  y  rings segments  seed amplitude
//     <file name as written in the XML>  limb   x0 y0 z0  x1 y1 z1  r0 r1  roundness  rings segments  seed amplitude
//     <file name as written in the XML>  membrane  ax ay az  bx by bz  cx cy cz  nu nv  seed amplitude
// Several lines with the same file name are the parts of ONE mesh (concatenated in table order).
// blob:  a sphere of radii (rx, ry, rz) around (cx, cy, cz) in a rings x segments latitude / longitude
//        grid (2 * segments * (rings - 1) triangles), displaced along its radius by a smooth pseudo-random
//        function of the direction, relative size `amplitude`;
// sheet: the rectangle [x0, x1] x [z0, z1] at height y in an nx x nz grid (2 nx nz triangles), displaced
//        in y by `amplitude` (absolute) times a smooth pseudo-random function;
// disc:  the annulus r0 <= r <= r1 around (cx, y, cz) in a rings x segments polar grid (2 rings segments
//        triangles; r0 = 0 gives a full disc whose innermost ring of triangles is degenerate-free because
//        the centre ring has radius r1 / (4 rings)), displaced in y like a sheet.
// limb:  a spindle around the segment p0 -> p1: circular cross-sections whose radius goes from r0 to r1
//        (linear) times sin(pi t)^(1/2) (roundness 1) or ^(1/4) (roundness 2: blunter ends), poles at both
//        ends, 2 * segments * (rings - 1) triangles, radius displaced by `amplitude` (relative);
// membrane: the triangle A B C as a fan of nu x nv quads from the apex A (rows start at 1 / (4 nu) of the
//        way, so that no triangle is degenerate), 2 nu nv triangles, displaced along the triangle's normal
//        by `amplitude` (absolute) times the smooth pseudo-random function.
// Vertices carry uv coordinates; normals come from the same post-processing an OBJ file without normals
// gets (mesh_postprocess.cpp); tangent frames are left to the commit (see Finish).  Everything is computed with + - * / sqrt and the
// restated sinf / cosf of glibc_libm.h: the same mesh on every host.
#include <cmath>
#include <sstream>
#include <stdexcept>

#include "../glibc_libm.h"
#include "standin_mesh.hpp"

namespace mcpt
{

namespace
{

uint32_t Mix(uint32_t v) // integer hash (lowbias32)
{
    v ^= v >> 16, v *= 0x7feb352du, v ^= v >> 15, v *= 0x846ca68bu, v ^= v >> 16;
    return v;
}

float Unit(uint32_t seed, uint32_t k) { return static_cast<float>(Mix(seed * 0x9e3779b9u + k) >> 8) * (1.0f / 16777216.0f); }

// Smooth pseudo-random function of a point, in [-1, 1]: a few sinusoids of random orientation and phase
// with frequencies 3 ... 40 and amplitudes ~ 1 / frequency.
struct Bumps
{
    static constexpr int kWaves = 12;
    float ax[kWaves], ay[kWaves], az[kWaves], phase[kWaves], weight[kWaves], norm;
    explicit Bumps(uint32_t seed)
    {
        float total = 0;
        for (int k = 0; k < kWaves; ++k)
        {
            const float freq = 3.0f + 37.0f * (static_cast<float>(k) / (kWaves - 1)) * (static_cast<float>(k) / (kWaves - 1));
            const float z = 2.0f * Unit(seed, 4 * k) - 1.0f, phi = 6.2831853f * Unit(seed, 4 * k + 1);
            const float s = std::sqrt(std::fmax(0.0f, 1.0f - z * z));
            ax[k] = freq * s * gl::cosf(phi), ay[k] = freq * z, az[k] = freq * s * gl::sinf(phi);
            phase[k] = 6.2831853f * Unit(seed, 4 * k + 2);
            weight[k] = 3.0f / freq;
            total += weight[k];
        }
        norm = 1.0f / total;
    }
    float operator()(float x, float y, float z) const
    {
        float v = 0;
        for (int k = 0; k < kWaves; ++k)
            v += weight[k] * gl::sinf(ax[k] * x + ay[k] * y + az[k] * z + phase[k]);
        return v * norm;
    }
};

// Smooth normals like an OBJ file without normals gets; NO per-vertex tangents: a stand-in is not an importer's
// output, so the commit gives it the reference's own per-triangle UV-derived frame (scene.cpp:63-80) — the pinned rule
// (SURVEY.md section 8c), the same one MCPT_MESH_TANGENTS=uv selects for files.
MeshData Finish(MeshData m)
{
    GenerateSmoothNormals(m);
    return m;
}

MeshData Blob(std::istringstream &in, const std::string &line)
{
    float c[3], r[3], amplitude;
    int rings, segments;
    uint32_t seed;
    if (!(in >> c[0] >> c[1] >> c[2] >> r[0] >> r[1] >> r[2] >> rings >> segments >> seed >> amplitude) || rings < 2 ||
        segments < 3 || rings > 8192 || segments > 8192)
        throw std::runtime_error("bad stand-in line '" + line + "'.");
    const Bumps bumps(seed);
    MeshData m;
    // vertices: one per pole, (rings - 1) x segments in between (no seam duplicates: uv wraps)
    auto put = [&](float dx, float dy, float dz, float u, float v)
    {
        const float k = 1.0f + amplitude * bumps(dx, dy, dz);
        m.positions.insert(m.positions.end(), {c[0] + r[0] * dx * k, c[1] + r[1] * dy * k, c[2] + r[2] * dz * k});
        m.texcoords.insert(m.texcoords.end(), {u, v});
    };
    put(0, 1, 0, 0.5f, 0.0f);
    for (int i = 1; i < rings; ++i)
    {
        const float theta = 3.14159265f * static_cast<float>(i) / static_cast<float>(rings);
        const float st = gl::sinf(theta), ct = gl::cosf(theta);
        for (int j = 0; j < segments; ++j)
        {
            const float phi = 6.2831853f * static_cast<float>(j) / static_cast<float>(segments);
            put(st * gl::cosf(phi), ct, st * gl::sinf(phi), static_cast<float>(j) / segments, static_cast<float>(i) / rings);
        }
    }
    put(0, -1, 0, 0.5f, 1.0f);
    const uint32_t south = static_cast<uint32_t>(m.positions.size() / 3 - 1);
    auto at = [&](int i, int j) { return 1u + static_cast<uint32_t>(i - 1) * segments + static_cast<uint32_t>(j % segments); };
    for (int j = 0; j < segments; ++j)
        m.indices.insert(m.indices.end(), {0u, at(1, j + 1), at(1, j)});
    for (int i = 1; i + 1 < rings; ++i)
        for (int j = 0; j < segments; ++j)
        {
            m.indices.insert(m.indices.end(), {at(i, j), at(i, j + 1), at(i + 1, j + 1)});
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j + 1), at(i + 1, j)});
        }
    for (int j = 0; j < segments; ++j)
        m.indices.insert(m.indices.end(), {south, at(rings - 1, j), at(rings - 1, j + 1)});
    return m;
}

MeshData Sheet(std::istringstream &in, const std::string &line)
{
    float x0, z0, x1, z1, y, amplitude;
    int nx, nz;
    uint32_t seed;
    if (!(in >> x0 >> z0 >> x1 >> z1 >> y >> nx >> nz >> seed >> amplitude) || nx < 1 || nz < 1 || nx > 8192 || nz > 8192)
        throw std::runtime_error("bad stand-in line '" + line + "'.");
    const Bumps bumps(seed);
    MeshData m;
    for (int i = 0; i <= nz; ++i)
        for (int j = 0; j <= nx; ++j)
        {
            const float u = static_cast<float>(j) / nx, v = static_cast<float>(i) / nz;
            const float x = x0 + (x1 - x0) * u, z = z0 + (z1 - z0) * v;
            m.positions.insert(m.positions.end(), {x, y + amplitude * bumps(u, 0.0f, v), z});
            m.texcoords.insert(m.texcoords.end(), {u, v});
        }
    auto at = [&](int i, int j) { return static_cast<uint32_t>(i * (nx + 1) + j); };
    for (int i = 0; i < nz; ++i)
        for (int j = 0; j < nx; ++j)
        {
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j), at(i + 1, j + 1)}); // facing +y
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j + 1), at(i, j + 1)});
        }
    return m;
}

MeshData Disc(std::istringstream &in, const std::string &line)
{
    float cx, cz, r0, r1, y, amplitude;
    int rings, segments;
    uint32_t seed;
    if (!(in >> cx >> cz >> r0 >> r1 >> y >> rings >> segments >> seed >> amplitude) || rings < 1 || segments < 3 ||
        rings > 8192 || segments > 8192 || !(r1 > r0) || r0 < 0)
        throw std::runtime_error("bad stand-in line '" + line + "'.");
    if (r0 == 0.0f)
        r0 = r1 / (4.0f * rings);
    const Bumps bumps(seed);
    MeshData m;
    for (int i = 0; i <= rings; ++i)
        for (int j = 0; j < segments; ++j)
        {
            const float v = static_cast<float>(i) / rings, u = static_cast<float>(j) / segments;
            const float r = r0 + (r1 - r0) * v, phi = 6.2831853f * u;
            const float dx = gl::cosf(phi), dz = gl::sinf(phi);
            m.positions.insert(m.positions.end(), {cx + r * dx, y + amplitude * bumps(v * dx, 0.0f, v * dz), cz + r * dz});
            m.texcoords.insert(m.texcoords.end(), {u, v});
        }
    auto at = [&](int i, int j) { return static_cast<uint32_t>(i * segments + (j % segments)); };
    for (int i = 0; i < rings; ++i)
        for (int j = 0; j < segments; ++j)
        {
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j + 1), at(i + 1, j)}); // facing +y
            m.indices.insert(m.indices.end(), {at(i, j), at(i, j + 1), at(i + 1, j + 1)});
        }
    return m;
}

MeshData Limb(std::istringstream &in, const std::string &line)
{
    float p0[3], p1[3], r0, r1, amplitude;
    int roundness, rings, segments;
    uint32_t seed;
    if (!(in >> p0[0] >> p0[1] >> p0[2] >> p1[0] >> p1[1] >> p1[2] >> r0 >> r1 >> roundness >> rings >> segments >> seed >>
          amplitude) ||
        rings < 2 || segments < 3 || rings > 8192 || segments > 8192 || roundness < 1 || roundness > 2)
        throw std::runtime_error("bad stand-in line '" + line + "'.");
    float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    const float length = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (!(length > 0))
        throw std::runtime_error("bad stand-in line '" + line + "'.");
    for (float &v : a)
        v /= length;
    // frame around the axis: u = a x h (h = +y, or +x for a vertical axis), v = a x u
    const bool steep = std::fabs(a[1]) >= 0.9f;
    const float h[3] = {steep ? 1.0f : 0.0f, steep ? 0.0f : 1.0f, 0.0f};
    float u[3] = {a[1] * h[2] - a[2] * h[1], a[2] * h[0] - a[0] * h[2], a[0] * h[1] - a[1] * h[0]};
    const float ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (float &v : u)
        v /= ul;
    const float w[3] = {a[1] * u[2] - a[2] * u[1], a[2] * u[0] - a[0] * u[2], a[0] * u[1] - a[1] * u[0]};
    const Bumps bumps(seed);
    MeshData m;
    auto put = [&](float t, float radius, float c, float s)
    {
        const float k = radius * (1.0f + amplitude * bumps(3.0f * t, c, s));
        for (int d = 0; d < 3; ++d)
            m.positions.push_back(p0[d] + a[d] * (length * t) + k * (c * u[d] + s * w[d]));
    };
    put(0.0f, 0.0f, 1.0f, 0.0f);
    m.texcoords.insert(m.texcoords.end(), {0.5f, 0.0f});
    for (int i = 1; i < rings; ++i)
    {
        const float t = static_cast<float>(i) / static_cast<float>(rings);
        float profile = std::sqrt(gl::sinf(3.14159265f * t));
        if (roundness == 2)
            profile = std::sqrt(profile);
        const float radius = (r0 + (r1 - r0) * t) * profile;
        for (int j = 0; j < segments; ++j)
        {
            const float phi = 6.2831853f * static_cast<float>(j) / static_cast<float>(segments);
            put(t, radius, gl::cosf(phi), gl::sinf(phi));
            m.texcoords.insert(m.texcoords.end(), {static_cast<float>(j) / segments, t});
        }
    }
    put(1.0f, 0.0f, 1.0f, 0.0f);
    m.texcoords.insert(m.texcoords.end(), {0.5f, 1.0f});
    const uint32_t last = static_cast<uint32_t>(m.positions.size() / 3 - 1);
    auto at = [&](int i, int j) { return 1u + static_cast<uint32_t>(i - 1) * segments + static_cast<uint32_t>(j % segments); };
    for (int j = 0; j < segments; ++j)
        m.indices.insert(m.indices.end(), {0u, at(1, j), at(1, j + 1)});
    for (int i = 1; i + 1 < rings; ++i)
        for (int j = 0; j < segments; ++j)
        {
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j + 1), at(i, j + 1)});
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j), at(i + 1, j + 1)});
        }
    for (int j = 0; j < segments; ++j)
        m.indices.insert(m.indices.end(), {last, at(rings - 1, j + 1), at(rings - 1, j)});
    return m;
}

MeshData Membrane(std::istringstream &in, const std::string &line)
{
    float A[3], B[3], C[3], amplitude;
    int nu, nv;
    uint32_t seed;
    if (!(in >> A[0] >> A[1] >> A[2] >> B[0] >> B[1] >> B[2] >> C[0] >> C[1] >> C[2] >> nu >> nv >> seed >> amplitude) || nu < 1 ||
        nv < 1 || nu > 8192 || nv > 8192)
        throw std::runtime_error("bad stand-in line '" + line + "'.");
    const float e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const float nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(nl > 0))
        throw std::runtime_error("bad stand-in line '" + line + "'.");
    for (float &v : n)
        v /= nl;
    const Bumps bumps(seed);
    MeshData m;
    const float u0 = 1.0f / (4.0f * nu);
    for (int i = 0; i <= nu; ++i)
        for (int j = 0; j <= nv; ++j)
        {
            const float u = u0 + (1.0f - u0) * static_cast<float>(i) / nu, v = static_cast<float>(j) / nv;
            const float lift = amplitude * bumps(u, 0.0f, v);
            for (int d = 0; d < 3; ++d)
                m.positions.push_back(A[d] + u * ((1.0f - v) * e1[d] + v * e2[d]) + lift * n[d]);
            m.texcoords.insert(m.texcoords.end(), {u, v});
        }
    auto at = [&](int i, int j) { return static_cast<uint32_t>(i * (nv + 1) + j); };
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nv; ++j)
        {
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j), at(i + 1, j + 1)});
            m.indices.insert(m.indices.end(), {at(i, j), at(i + 1, j + 1), at(i, j + 1)});
        }
    return m;
}

// Parts of one mesh: indices shifted, attribute arrays concatenated (before normals / tangents are made).
void Append(MeshData &to, MeshData &&part)
{
    const uint32_t base = static_cast<uint32_t>(to.positions.size() / 3);
    to.positions.insert(to.positions.end(), part.positions.begin(), part.positions.end());
    to.texcoords.insert(to.texcoords.end(), part.texcoords.begin(), part.texcoords.end());
    for (uint32_t i : part.indices)
        to.indices.push_back(base + i);
}

} // namespace

StandinTable::StandinTable(const std::string &text)
{
    std::istringstream lines(text);
    std::string line;
    while (std::getline(lines, line))
    {
        std::istringstream in(line);
        std::string name, kind;
        if (!(in >> name) || name[0] == '#')
            continue;
        if (!(in >> kind) || (kind != "blob" && kind != "sheet" && kind != "disc" && kind != "limb" && kind != "membrane"))
            throw std::runtime_error("bad stand-in line '" + line + "'.");
        lines_[name].push_back(line);
    }
}

bool StandinTable::Has(const std::string &name) const { return lines_.count(name) != 0; }

MeshData StandinTable::Build(const std::string &name) const
{
    MeshData mesh;
    for (const std::string &line : lines_.at(name))
    {
        std::istringstream in(line);
        std::string skip, kind;
        in >> skip >> kind;
        MeshData part = kind == "blob"    ? Blob(in, line)
                        : kind == "sheet" ? Sheet(in, line)
                        : kind == "disc"  ? Disc(in, line)
                        : kind == "limb"  ? Limb(in, line)
                                          : Membrane(in, line);
        if (mesh.positions.empty())
            mesh = std::move(part);
        else
            Append(mesh, std::move(part));
    }
    return Finish(std::move(mesh));
}

} // namespace mcpt
