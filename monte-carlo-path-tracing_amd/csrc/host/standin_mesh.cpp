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
// blob:  a sphere of radii (rx, ry, rz) around (cx, cy, cz) in a rings x segments latitude / longitude
//        grid (2 * segments * (rings - 1) triangles), displaced along its radius by a smooth pseudo-random
//        function of the direction, relative size `amplitude`;
// sheet: the rectangle [x0, x1] x [z0, z1] at height y in an nx x nz grid (2 nx nz triangles), displaced
//        in y by `amplitude` (absolute) times a smooth pseudo-random function;
// disc:  the annulus r0 <= r <= r1 around (cx, y, cz) in a rings x segments polar grid (2 rings segments
//        triangles; r0 = 0 gives a full disc whose innermost ring of triangles is degenerate-free because
//        the centre ring has radius r1 / (4 rings)), displaced in y like a sheet.
// Vertices carry uv coordinates; normals and tangent frames come from the same post-processing an
// OBJ file without normals gets (mesh_postprocess.cpp).  Everything is computed with + - * / sqrt and the
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

MeshData Finish(MeshData m)
{
    GenerateSmoothNormals(m);
    CalcTangentSpace(m);
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
    return Finish(std::move(m));
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
    return Finish(std::move(m));
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
    return Finish(std::move(m));
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
        if (!(in >> kind) || (kind != "blob" && kind != "sheet" && kind != "disc"))
            throw std::runtime_error("bad stand-in line '" + line + "'.");
        lines_[name] = line;
    }
}

bool StandinTable::Has(const std::string &name) const { return lines_.count(name) != 0; }

MeshData StandinTable::Build(const std::string &name) const
{
    const std::string &line = lines_.at(name);
    std::istringstream in(line);
    std::string skip, kind;
    in >> skip >> kind;
    return kind == "blob" ? Blob(in, line) : (kind == "sheet" ? Sheet(in, line) : Disc(in, line));
}

} // namespace mcpt
