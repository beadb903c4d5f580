// Asset readers of the scene front end: Wavefront OBJ, Mitsuba ".serialized"
// meshes (zlib), and float images for bitmap textures / environment maps
// (PFM; OpenEXR scanline files with NONE / ZIPS / ZIP / PIZ compression).
//
// Reference counterparts: src/parser/model_loader.cpp (assimp for OBJ,
// :426-504 for .serialized) and src/utils/image_io.cpp:55-158 (tinyexr, stb).
// Neither assimp nor tinyexr exists here, so these are written from the file
// format definitions.  Mesh records come out the way the reference hands them
// to the renderer: positions / normals / texcoords / tangents / bitangents per vertex +
// index triples; generated normals and the tangent frames restate assimp's
// post-processing steps (mesh_postprocess.cpp; unpinned — SURVEY.md §8c).
#include "asset_io.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include <zlib.h>

// MCPT_MESH_TANGENTS=uv: OBJ / PLY meshes are handed over WITHOUT per-vertex tangents and bitangents, which selects the
// reference's own per-triangle UV-derived frame in the commit (scene.cpp:63-80) — the pinned alternative to the restatement
// of the importer's CalcTangentSpace (mesh_postprocess.cpp: unpinned, there is no assimp here to compare with; SURVEY.md
// section 8c names this pin).  A maintainer who binds the library behind the reference's own assimp loader passes assimp's
// tangents and needs neither.
static bool UvDerivedTangentsOnly()
{
    const char *e = std::getenv("MCPT_MESH_TANGENTS");
    return e && e[0] == 'u';
}

namespace mcpt
{
namespace
{

std::vector<uint8_t> ReadAll(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw std::runtime_error("read file '" + path + "' failed.");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

std::string SuffixOf(const std::string &path)
{
    const size_t dot = path.find_last_of('.');
    std::string s = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (char &c : s)
        c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}

} // namespace

// ---------------------------------------------------------------------------
// PLY (ascii and binary_little_endian): x y z [nx ny nz] [s t | u v] per vertex,
// faces as index lists (fans).  The reference imports PLY through assimp with
// aiProcess_GenSmoothNormals (model_loader.cpp:506-529); without stored normals the
// vertex normals here are the normalised sums of the (area-weighted) face normals of the
// faces sharing the vertex — assimp's rule for vertices that share an index; assimp also
// merges vertices at equal positions and limits the smoothing angle, which is not
// reproduced (same caveat as for OBJ tangents).
// ---------------------------------------------------------------------------
MeshData LoadPly(const std::string &path, bool face_normals)
{
    const std::vector<uint8_t> f = ReadAll(path);
    size_t at = 0;
    auto line = [&]()
    {
        std::string s;
        while (at < f.size() && f[at] != '\n')
            s += static_cast<char>(f[at++]);
        ++at;
        if (!s.empty() && s.back() == '\r')
            s.pop_back();
        return s;
    };
    if (line() != "ply")
        throw std::runtime_error("not a PLY file: '" + path + "'.");
    struct Property
    {
        std::string name, type, count_type; // count_type non-empty: list
    };
    struct Element
    {
        std::string name;
        size_t count = 0;
        std::vector<Property> props;
    };
    std::vector<Element> elements;
    bool binary = false;
    for (;;)
    {
        if (at >= f.size())
            throw std::runtime_error("truncated PLY header in '" + path + "'.");
        std::istringstream ss(line());
        std::string word;
        ss >> word;
        if (word == "end_header")
            break;
        if (word == "format")
        {
            ss >> word;
            if (word == "binary_little_endian")
                binary = true;
            else if (word != "ascii")
                throw std::runtime_error("unsupported PLY format '" + word + "' in '" + path + "'.");
        }
        else if (word == "element")
        {
            Element e;
            ss >> e.name >> e.count;
            elements.push_back(e);
        }
        else if (word == "property" && !elements.empty())
        {
            Property p;
            ss >> p.type;
            if (p.type == "list")
                ss >> p.count_type >> p.type;
            ss >> p.name;
            elements.back().props.push_back(p);
        }
    }
    auto type_size = [&](const std::string &t) -> size_t
    {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8")
            return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16")
            return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32")
            return 4;
        if (t == "double" || t == "float64")
            return 8;
        throw std::runtime_error("unknown PLY type '" + t + "' in '" + path + "'.");
    };
    std::istringstream text;
    if (!binary)
        text.str(std::string(reinterpret_cast<const char *>(f.data()) + at, f.size() - at));
    auto read_number = [&](const std::string &t) -> double
    {
        if (!binary)
        {
            double v = 0;
            if (!(text >> v))
                throw std::runtime_error("truncated PLY body in '" + path + "'.");
            return v;
        }
        const size_t n = type_size(t);
        if (at + n > f.size())
            throw std::runtime_error("truncated PLY body in '" + path + "'.");
        double v = 0;
        const uint8_t *p = &f[at];
        at += n;
        if (t == "float" || t == "float32")
        {
            float x;
            std::memcpy(&x, p, 4);
            v = x;
        }
        else if (t == "double" || t == "float64")
            std::memcpy(&v, p, 8);
        else if (t == "char" || t == "int8")
            v = static_cast<int8_t>(p[0]);
        else if (t == "uchar" || t == "uint8")
            v = p[0];
        else if (t == "short" || t == "int16")
        {
            int16_t x;
            std::memcpy(&x, p, 2);
            v = x;
        }
        else if (t == "ushort" || t == "uint16")
        {
            uint16_t x;
            std::memcpy(&x, p, 2);
            v = x;
        }
        else if (t == "int" || t == "int32")
        {
            int32_t x;
            std::memcpy(&x, p, 4);
            v = x;
        }
        else
        {
            uint32_t x;
            std::memcpy(&x, p, 4);
            v = x;
        }
        return v;
    };
    MeshData m;
    bool has_normals = false, has_uv = false;
    for (const Element &e : elements)
    {
        int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
        for (size_t k = 0; k < e.props.size(); ++k)
        {
            const std::string &n = e.props[k].name;
            const int i = static_cast<int>(k);
            if (n == "x") ix = i; else if (n == "y") iy = i; else if (n == "z") iz = i;
            else if (n == "nx") inx = i; else if (n == "ny") iny = i; else if (n == "nz") inz = i;
            else if (n == "s" || n == "u" || n == "texture_u") iu = i;
            else if (n == "t" || n == "v" || n == "texture_v") iv = i;
        }
        if (e.name == "vertex")
        {
            if (ix < 0 || iy < 0 || iz < 0)
                throw std::runtime_error("PLY vertices without x y z in '" + path + "'.");
            has_normals = inx >= 0 && iny >= 0 && inz >= 0, has_uv = iu >= 0 && iv >= 0;
        }
        std::vector<double> row;
        for (size_t r = 0; r < e.count; ++r)
        {
            row.assign(e.props.size(), 0.0);
            std::vector<uint32_t> list;
            for (size_t k = 0; k < e.props.size(); ++k)
            {
                const Property &p = e.props[k];
                if (p.count_type.empty())
                    row[k] = read_number(p.type);
                else
                {
                    const size_t n = static_cast<size_t>(read_number(p.count_type));
                    std::vector<uint32_t> values(n);
                    for (size_t j = 0; j < n; ++j)
                        values[j] = static_cast<uint32_t>(read_number(p.type));
                    if (e.name == "face" && (p.name == "vertex_indices" || p.name == "vertex_index"))
                        list = values;
                }
            }
            if (e.name == "vertex")
            {
                m.positions.insert(m.positions.end(), {float(row[ix]), float(row[iy]), float(row[iz])});
                if (has_normals)
                    m.normals.insert(m.normals.end(), {float(row[inx]), float(row[iny]), float(row[inz])});
                if (has_uv)
                    m.texcoords.insert(m.texcoords.end(), {float(row[iu]), float(row[iv])});
            }
            else if (e.name == "face")
                for (size_t j = 1; j + 1 < list.size(); ++j)
                    m.indices.insert(m.indices.end(), {list[0], list[j], list[j + 1]});
        }
    }
    const size_t n_vert = m.positions.size() / 3;
    for (const uint32_t i : m.indices)
        if (i >= n_vert)
            throw std::runtime_error("vertex index out of range in '" + path + "'.");
    if (face_normals)
        m.normals.clear();
    else if (!has_normals)
    {
        std::vector<double> sum(3 * n_vert, 0.0);
        for (size_t t = 0; t + 2 < m.indices.size(); t += 3)
        {
            const float *a = &m.positions[3 * m.indices[t]], *b = &m.positions[3 * m.indices[t + 1]],
                        *c = &m.positions[3 * m.indices[t + 2]];
            const double e1[3] = {double(b[0]) - a[0], double(b[1]) - a[1], double(b[2]) - a[2]},
                         e2[3] = {double(c[0]) - a[0], double(c[1]) - a[1], double(c[2]) - a[2]};
            const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j)
                    sum[3 * m.indices[t + k] + j] += n[j];
        }
        m.normals.resize(3 * n_vert);
        for (size_t v = 0; v < n_vert; ++v)
        {
            const double len = std::sqrt(sum[3 * v] * sum[3 * v] + sum[3 * v + 1] * sum[3 * v + 1] + sum[3 * v + 2] * sum[3 * v + 2]);
            for (int j = 0; j < 3; ++j)
                m.normals[3 * v + j] = len > 0 ? static_cast<float>(sum[3 * v + j] / len) : (j == 1 ? 1.0f : 0.0f);
        }
    }
    if (!face_normals && !UvDerivedTangentsOnly())
        CalcTangentSpace(m); // (with face_normals a PLY without stored normals has none to build the frame on)
    return m;
}

// ---------------------------------------------------------------------------
// OBJ.  Like the reference's assimp import without aiProcess_JoinIdenticalVertices
// (model_loader.cpp:512-517): one vertex per face corner, faces triangulated as
// fans, v texture coordinate flipped when asked (aiProcess_FlipUVs).
// ---------------------------------------------------------------------------
MeshData LoadObj(const std::string &path, bool flip_texcoords, bool face_normals)
{
    std::ifstream f(path);
    if (!f)
        throw std::runtime_error("read file '" + path + "' failed.");
    std::vector<float> v, vt, vn;
    MeshData m;
    bool all_have_uv = true, all_have_normal = true, any_face = false;
    std::string line;
    auto resolve = [](long idx, size_t count) -> long
    {
        if (idx > 0)
            return idx - 1;
        if (idx < 0)
            return static_cast<long>(count) + idx;
        return -1;
    };
    struct Corner
    {
        long p, t, n;
    };
    std::vector<Corner> corners;
    while (std::getline(f, line))
    {
        if (line.size() < 2)
            continue;
        const char *s = line.c_str();
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t'))
        {
            float x = 0, y = 0, z = 0;
            std::sscanf(s + 2, "%f %f %f", &x, &y, &z);
            v.insert(v.end(), {x, y, z});
        }
        else if (s[0] == 'v' && s[1] == 't')
        {
            float a = 0, b = 0;
            std::sscanf(s + 3, "%f %f", &a, &b);
            vt.insert(vt.end(), {a, b});
        }
        else if (s[0] == 'v' && s[1] == 'n')
        {
            float x = 0, y = 0, z = 0;
            std::sscanf(s + 3, "%f %f %f", &x, &y, &z);
            vn.insert(vn.end(), {x, y, z});
        }
        else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t'))
        {
            corners.clear();
            std::istringstream ss(line.substr(2));
            std::string tok;
            while (ss >> tok)
            {
                Corner c{0, 0, 0};
                long a = 0, b = 0, d = 0;
                if (std::sscanf(tok.c_str(), "%ld/%ld/%ld", &a, &b, &d) == 3)
                    c = {a, b, d};
                else if (std::sscanf(tok.c_str(), "%ld//%ld", &a, &d) == 2)
                    c = {a, 0, d};
                else if (std::sscanf(tok.c_str(), "%ld/%ld", &a, &b) == 2)
                    c = {a, b, 0};
                else if (std::sscanf(tok.c_str(), "%ld", &a) == 1)
                    c = {a, 0, 0};
                else
                    throw std::runtime_error("malformed face in '" + path + "'.");
                corners.push_back(c);
            }
            for (size_t k = 1; k + 1 < corners.size(); ++k)
            {
                const Corner tri[3] = {corners[0], corners[k], corners[k + 1]};
                for (const Corner &c : tri)
                {
                    const long p = resolve(c.p, v.size() / 3), t = resolve(c.t, vt.size() / 2),
                               n = resolve(c.n, vn.size() / 3);
                    if (p < 0 || static_cast<size_t>(p) >= v.size() / 3)
                        throw std::runtime_error("vertex index out of range in '" + path + "'.");
                    m.positions.insert(m.positions.end(), {v[3 * p], v[3 * p + 1], v[3 * p + 2]});
                    if (t >= 0 && static_cast<size_t>(t) < vt.size() / 2)
                        m.texcoords.insert(m.texcoords.end(),
                                           {vt[2 * t], flip_texcoords ? 1.0f - vt[2 * t + 1] : vt[2 * t + 1]});
                    else
                        all_have_uv = false;
                    if (n >= 0 && static_cast<size_t>(n) < vn.size() / 3)
                        m.normals.insert(m.normals.end(), {vn[3 * n], vn[3 * n + 1], vn[3 * n + 2]});
                    else
                        all_have_normal = false;
                    m.indices.push_back(static_cast<uint32_t>(m.indices.size()));
                }
                any_face = true;
            }
        }
    }
    if (!any_face)
        throw std::runtime_error("no faces in '" + path + "'.");
    if (!all_have_uv)
        m.texcoords.clear();
    if (!all_have_normal)
        m.normals.clear();
    // the importer steps the reference asks for (model_loader.cpp:512-517), in assimp's order
    if (m.normals.empty() && !face_normals)
        GenerateSmoothNormals(m);
    if (!UvDerivedTangentsOnly())
        CalcTangentSpace(m); // from the file's or the generated normals; nothing without normals or texcoords
    if (face_normals)
        m.normals.clear(); // not handed over (model_loader.cpp:362); flat normals come from the commit (scene.cpp:51-56)
    return m;
}

// ---------------------------------------------------------------------------
// Mitsuba .serialized (format 0x041C, versions 3 and 4): header, zlib stream of
// {flags, [name], vertex count, triangle count, positions, [normals],
// [texcoords], [colors], indices}; the offset table of the sub-meshes sits at
// the end of the file.  model_loader.cpp:426-504.
// ---------------------------------------------------------------------------
MeshData LoadSerialized(const std::string &path, int shape_index)
{
    const std::vector<uint8_t> file = ReadAll(path);
    auto u16 = [&](size_t at)
    {
        if (at + 2 > file.size())
            throw std::runtime_error("read less data than expected from file '" + path + "'.");
        return static_cast<uint16_t>(file[at] | (file[at + 1] << 8));
    };
    auto u32 = [&](size_t at)
    {
        if (at + 4 > file.size())
            throw std::runtime_error("read less data than expected from file '" + path + "'.");
        uint32_t x;
        std::memcpy(&x, &file[at], 4);
        return x;
    };
    auto u64 = [&](size_t at)
    {
        if (at + 8 > file.size())
            throw std::runtime_error("read less data than expected from file '" + path + "'.");
        uint64_t x;
        std::memcpy(&x, &file[at], 8);
        return x;
    };
    if (u16(0) != 0x041C)
        throw std::runtime_error("invalid file format for '" + path + "'.");
    const uint16_t version = u16(2);
    if (version != 3 && version != 4)
        throw std::runtime_error("invalid file version for '" + path + "'.");
    size_t offset = 0;
    if (shape_index != 0)
    {
        const uint32_t count = u32(file.size() - 4);
        if (shape_index < 0 || shape_index > static_cast<int>(count))
            throw std::runtime_error("unable to unserialize mesh, shape index is out of range for file '" + path + "'.");
        if (version == 4)
            offset = static_cast<size_t>(u64(file.size() - 8ull * (count - shape_index) - 4));
        else
            offset = u32(file.size() - 4ull * (count - shape_index + 1));
        if (u16(offset) != 0x041C)
            throw std::runtime_error("invalid sub-mesh header in '" + path + "'.");
    }
    // inflate everything that follows the 4-byte sub-mesh header, on demand
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15) != Z_OK)
        throw std::runtime_error("cannot initialize ZLIB.");
    zs.next_in = const_cast<Bytef *>(file.data() + offset + 4);
    zs.avail_in = static_cast<uInt>(std::min<size_t>(file.size() - offset - 4, 0xFFFFFFFFu));
    auto read = [&](void *dst, size_t bytes)
    {
        zs.next_out = static_cast<Bytef *>(dst);
        while (bytes > 0)
        {
            zs.avail_out = static_cast<uInt>(std::min<size_t>(bytes, 1u << 30));
            const uInt want = zs.avail_out;
            const int rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END)
            {
                inflateEnd(&zs);
                throw std::runtime_error("inflate(): data error for file '" + path + "'.");
            }
            bytes -= want - zs.avail_out;
            if (bytes > 0 && rc == Z_STREAM_END)
            {
                inflateEnd(&zs);
                throw std::runtime_error("inflate(): attempting to read past the end of the stream for file '" + path + "'.");
            }
        }
    };
    uint32_t flags;
    read(&flags, 4);
    if (version == 4)
    {
        char c;
        do
            read(&c, 1);
        while (c != 0);
    }
    uint64_t n_vert, n_tri;
    read(&n_vert, 8);
    read(&n_tri, 8);
    if (n_vert > (1ull << 31) || n_tri > (1ull << 31))
    {
        inflateEnd(&zs);
        throw std::runtime_error("implausible mesh size in '" + path + "'.");
    }
    const bool dbl = (flags & 0x2000u) != 0;
    auto read_floats = [&](std::vector<float> &dst, size_t count)
    {
        dst.resize(count);
        if (dbl)
        {
            std::vector<double> tmp(count);
            read(tmp.data(), 8 * count);
            for (size_t i = 0; i < count; ++i)
                dst[i] = static_cast<float>(tmp[i]);
        }
        else
            read(dst.data(), 4 * count);
    };
    MeshData m;
    read_floats(m.positions, 3 * n_vert);
    if (flags & 0x0001u)
        read_floats(m.normals, 3 * n_vert);
    if (flags & 0x0002u)
        read_floats(m.texcoords, 2 * n_vert);
    if (flags & 0x0008u)
    {
        std::vector<float> colors;
        read_floats(colors, 3 * n_vert);
    }
    m.indices.resize(3 * n_tri);
    read(m.indices.data(), 12 * n_tri);
    inflateEnd(&zs);
    return m;
}

// ---------------------------------------------------------------------------
// Float images
// ---------------------------------------------------------------------------
namespace
{

float HalfToFloat(uint16_t h)
{
    const uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0)
    {
        if (man == 0)
            bits = sign;
        else
        {
            int e = -1;
            uint32_t m2 = man;
            do
            {
                ++e;
                m2 <<= 1;
            } while (!(m2 & 0x400u));
            bits = sign | ((127 - 15 - e) << 23) | ((m2 & 0x3FFu) << 13);
        }
    }
    else if (exp == 31)
        bits = sign | 0x7F800000u | (man << 13);
    else
        bits = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

ImageData LoadPfm(const std::string &path)
{
    const std::vector<uint8_t> file = ReadAll(path);
    const char *p = reinterpret_cast<const char *>(file.data());
    int w = 0, h = 0, consumed = 0;
    float scale = 0;
    char tag[3] = {0, 0, 0};
    if (std::sscanf(p, "%2s %d %d %f%n", tag, &w, &h, &scale, &consumed) != 4 || w <= 0 || h <= 0)
        throw std::runtime_error("malformed PFM header in '" + path + "'.");
    const int channels = std::strcmp(tag, "PF") == 0 ? 3 : (std::strcmp(tag, "Pf") == 0 ? 1 : 0);
    if (!channels)
        throw std::runtime_error("unsupported PFM type in '" + path + "'.");
    const size_t data_at = static_cast<size_t>(consumed) + 1;
    const size_t n = static_cast<size_t>(w) * h * channels;
    if (data_at + 4 * n > file.size())
        throw std::runtime_error("truncated PFM '" + path + "'.");
    ImageData img;
    img.width = w, img.height = h, img.channel = channels;
    img.data.resize(n);
    const bool little = scale < 0;
    for (int y = 0; y < h; ++y) // PFM stores the bottom row first
        for (int x = 0; x < w * channels; ++x)
        {
            uint8_t b[4];
            std::memcpy(b, &file[data_at + 4 * (static_cast<size_t>(h - 1 - y) * w * channels + x)], 4);
            if (!little)
                std::swap(b[0], b[3]), std::swap(b[1], b[2]);
            std::memcpy(&img.data[static_cast<size_t>(y) * w * channels + x], b, 4);
        }
    return img;
}

// OpenEXR, single-part scanline, compression NONE / ZIPS / ZIP / PIZ (exr_piz.cpp), HALF or FLOAT
// channels.  Output: RGBA float with alpha 1 when absent ("channel = 4", as the
// reference's tinyexr path returns, image_io.cpp:75-98).
ImageData LoadExr(const std::string &path)
{
    const std::vector<uint8_t> f = ReadAll(path);
    size_t at = 0;
    auto need = [&](size_t n)
    {
        if (at + n > f.size())
            throw std::runtime_error("truncated EXR '" + path + "'.");
    };
    auto rd32 = [&]()
    {
        need(4);
        uint32_t x;
        std::memcpy(&x, &f[at], 4);
        at += 4;
        return x;
    };
    auto rdstr = [&]()
    {
        std::string s;
        for (;;)
        {
            need(1);
            const char c = static_cast<char>(f[at++]);
            if (!c)
                break;
            s += c;
        }
        return s;
    };
    if (rd32() != 20000630u)
        throw std::runtime_error("not an OpenEXR file: '" + path + "'.");
    const uint32_t version = rd32();
    if (version & 0x1E00u)
        throw std::runtime_error("tiled / deep / multi-part EXR is not supported: '" + path + "'.");
    struct Channel
    {
        std::string name;
        int type;
    };
    std::vector<Channel> channels;
    int compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1, line_order = 0;
    for (;;)
    {
        const std::string name = rdstr();
        if (name.empty())
            break;
        const std::string type = rdstr();
        const uint32_t size = rd32();
        need(size);
        const size_t end = at + size;
        if (name == "channels")
        {
            while (at < end && f[at] != 0)
            {
                Channel c;
                c.name = rdstr();
                c.type = static_cast<int>(rd32());
                at += 4; // pLinear + reserved
                const uint32_t xs = rd32(), ys = rd32();
                if (xs != 1 || ys != 1)
                    throw std::runtime_error("subsampled EXR channels are not supported: '" + path + "'.");
                channels.push_back(c);
            }
        }
        else if (name == "compression")
            compression = f[at];
        else if (name == "dataWindow")
        {
            int v[4];
            std::memcpy(v, &f[at], 16);
            x0 = v[0], y0 = v[1], x1 = v[2], y1 = v[3];
        }
        else if (name == "lineOrder")
            line_order = f[at];
        at = end;
    }
    (void)line_order;
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    if (w <= 0 || h <= 0 || channels.empty())
        throw std::runtime_error("malformed EXR header in '" + path + "'.");
    int lines_per_block;
    switch (compression)
    {
    case 0: // NONE
    case 2: // ZIPS
        lines_per_block = 1;
        break;
    case 3: // ZIP
        lines_per_block = 16;
        break;
    case 4: // PIZ
        lines_per_block = 32;
        break;
    default:
        throw std::runtime_error("EXR compression " + std::to_string(compression) +
                                 " (only NONE / ZIPS / ZIP / PIZ are supported) in '" + path + "'.");
    }
    size_t bytes_per_pixel = 0;
    for (const Channel &c : channels)
        bytes_per_pixel += c.type == 1 ? 2 : 4;
    const int n_blocks = (h + lines_per_block - 1) / lines_per_block;
    std::vector<uint64_t> offsets(n_blocks);
    need(8ull * n_blocks);
    std::memcpy(offsets.data(), &f[at], 8ull * n_blocks);
    ImageData img;
    img.width = w, img.height = h, img.channel = 4;
    img.data.assign(static_cast<size_t>(w) * h * 4, 0.0f);
    for (size_t i = 3; i < img.data.size(); i += 4)
        img.data[i] = 1.0f;
    std::vector<uint8_t> raw, tmp;
    for (int b = 0; b < n_blocks; ++b)
    {
        at = static_cast<size_t>(offsets[b]);
        const int y_first = static_cast<int>(rd32()) - y0;
        const uint32_t packed = rd32();
        need(packed);
        const int lines = std::min(lines_per_block, h - y_first);
        const size_t expect = bytes_per_pixel * w * lines;
        raw.resize(expect);
        if (compression == 0 || packed == expect)
            std::memcpy(raw.data(), &f[at], std::min<size_t>(packed, expect));
        else if (compression == 4)
        {
            std::vector<int> words;
            for (const Channel &c : channels)
                words.push_back(c.type == 1 ? 1 : 2);
            DecodePizBlock(&f[at], packed, words, w, lines, raw.data());
        }
        else
        {
            tmp.resize(expect);
            uLongf out_len = static_cast<uLongf>(expect);
            if (uncompress(tmp.data(), &out_len, &f[at], packed) != Z_OK || out_len != expect)
                throw std::runtime_error("corrupt ZIP block in EXR '" + path + "'.");
            // undo the predictor, then the even/odd byte interleave
            for (size_t i = 1; i < expect; ++i)
                tmp[i] = static_cast<uint8_t>(tmp[i - 1] + tmp[i] - 128);
            const size_t half = (expect + 1) / 2;
            for (size_t i = 0; i < expect; ++i)
                raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
        }
        // block layout: per scanline, per channel (alphabetical), w samples
        size_t cursor = 0;
        for (int l = 0; l < lines; ++l)
        {
            const int y = y_first + l;
            for (const Channel &c : channels)
            {
                int slot = -1;
                if (c.name == "R")
                    slot = 0;
                else if (c.name == "G")
                    slot = 1;
                else if (c.name == "B")
                    slot = 2;
                else if (c.name == "A")
                    slot = 3;
                else if (c.name == "Y")
                    slot = 4; // luminance-only image: replicate
                for (int x = 0; x < w; ++x)
                {
                    float v;
                    if (c.type == 1)
                    {
                        uint16_t hv;
                        std::memcpy(&hv, &raw[cursor], 2);
                        cursor += 2;
                        v = HalfToFloat(hv);
                    }
                    else if (c.type == 2)
                    {
                        std::memcpy(&v, &raw[cursor], 4);
                        cursor += 4;
                    }
                    else
                    {
                        uint32_t u;
                        std::memcpy(&u, &raw[cursor], 4);
                        cursor += 4;
                        v = static_cast<float>(u);
                    }
                    float *px = &img.data[(static_cast<size_t>(y) * w + x) * 4];
                    if (slot >= 0 && slot <= 3)
                        px[slot] = v;
                    else if (slot == 4)
                        px[0] = px[1] = px[2] = v;
                }
            }
        }
    }
    return img;
}

} // namespace

ImageData LoadFloatImage(const std::string &path, float gamma)
{
    const std::string suffix = SuffixOf(path);
    auto srgb_to_linear = [](float v) // image_io.cpp:113-118, 137-142
    { return v <= 0.04045f ? v / 12.92f : static_cast<float>(std::pow((v + 0.055f) / 1.055f, 2.4f)); };
    ImageData img;
    if (suffix == "exr")
    {
        img = LoadExr(path);
        if (gamma != 0.0f)
        {
            // image_io.cpp:91-96: applied while the channel count still reads 1, i.e. to the
            // first width*height floats of the RGBA data
            const size_t n = static_cast<size_t>(img.width) * img.height;
            for (size_t i = 0; i < n; ++i)
                img.data[i] = std::pow(img.data[i], gamma);
        }
    }
    else if (suffix == "pfm")
    {
        img = LoadPfm(path);
        if (gamma != 0.0f)
            for (float &v : img.data)
                v = std::pow(v, gamma);
    }
    else if (suffix == "hdr" || suffix == "pic")
    {
        img = LoadRadianceHdr(path);
        if (gamma == -1.0f)
            for (float &v : img.data)
                v = srgb_to_linear(v);
        else if (gamma != 0.0f)
            for (float &v : img.data)
                v = std::pow(v, gamma);
    }
    else if (suffix == "png" || suffix == "jpg" || suffix == "jpeg")
    {
        std::vector<uint8_t> px;
        if (suffix == "png")
            LoadPng8(path, img.width, img.height, img.channel, px);
        else
            LoadJpeg8(path, img.width, img.height, img.channel, px);
        img.data.resize(px.size());
        for (size_t i = 0; i < px.size(); ++i)
        {
            const float v = static_cast<int>(px[i]) / 255.0f;
            img.data[i] = (gamma == 0.0f || gamma == -1.0f) ? srgb_to_linear(v) : std::pow(v, gamma);
        }
    }
    else
    {
        throw std::runtime_error("unsupport input image format for image '" + path +
                                 "' (supported: .exr, .pfm, .hdr, .png, .jpg).");
    }
    return img;
}

} // namespace mcpt
