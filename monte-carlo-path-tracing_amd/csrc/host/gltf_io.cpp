// glTF 2.0 meshes (`<shape type="gltf">`, reference src/parser/parser.cpp:1163-1191): the reference hands the file to
// assimp (model_loader.cpp:506-530: Triangulate | GenUVCoords | CalcTangentSpace, GenSmoothNormals unless face_normals, no
// FlipUVs for this type) and flattens the imported node tree with ProcessAssimpNode (model_loader.cpp:335-419).  assimp is
// not in this image, so — like the OBJ / PLY readers next door (asset_io.cpp) — this is a restatement of
// what that import delivers, UNPINNED (nothing here can be compared with assimp):
//   * .gltf (JSON + external or base64 buffers) and .glb (binary container); the default scene's node tree, depth first;
//   * every triangle primitive (mode 4; strips 5 and fans 6 are expanded) of every mesh a node names becomes one imported
//     mesh: POSITION, NORMAL, TEXCOORD_0 (v flipped: assimp's glTF importer stores 1 - v) and the index accessor, indexed
//     vertices kept as they are (no vertex joining is requested);
//   * node TRANSFORMS ARE IGNORED, as ProcessAssimpNode ignores aiNode::mTransformation (reference quirk: a glTF whose
//     nodes carry transforms renders untransformed there too; the <transform> of the shape applies as usual);
//   * the flattening's index offset is the reference's own: a node's meshes are appended with the offset its parent handed
//     down, which is the number of TRIANGLES (not vertices) gathered before the node (model_loader.cpp:345-347, 395-396) —
//     files with one mesh (the usual export of one object) are unaffected, files with several come out as garbled as there;
//   * then the importer steps of mesh_postprocess.cpp in assimp's order (smooth normals if the file has none, tangent frames
//     from normals and texture coordinates).
#include "asset_io.hpp"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mcpt
{

namespace
{

// ---- a small JSON reader (RFC 8259; enough for glTF: no streaming, numbers as double) ----
struct Json
{
    enum Kind
    {
        kNull,
        kBool,
        kNumber,
        kString,
        kArray,
        kObject
    } kind = kNull;
    bool boolean = false;
    double number = 0.0;
    std::string string;
    std::vector<Json> array;
    std::vector<std::pair<std::string, Json>> object;

    const Json *Find(const char *key) const
    {
        for (const auto &kv : object)
            if (kv.first == key)
                return &kv.second;
        return nullptr;
    }
    double Number(const char *key, double fallback) const
    {
        const Json *v = Find(key);
        return v && v->kind == kNumber ? v->number : fallback;
    }
    long Index(const char *key) const // a non-negative integer member, -1 if absent
    {
        const Json *v = Find(key);
        return v && v->kind == kNumber && v->number >= 0 && v->number <= 9.0e15 ? static_cast<long>(v->number) : -1; // (beyond that: not an index)
    }
    std::string String(const char *key) const
    {
        const Json *v = Find(key);
        return v && v->kind == kString ? v->string : std::string();
    }
};

class JsonReader
{
public:
    JsonReader(const char *begin, const char *end, const std::string &what) : p_(begin), end_(end), what_(what) {}
    Json Parse()
    {
        Json v = Value(0);
        Space();
        if (p_ != end_)
            Fail("trailing characters");
        return v;
    }

private:
    [[noreturn]] void Fail(const char *why) const { throw std::runtime_error("malformed JSON in '" + what_ + "': " + why + "."); }
    void Space()
    {
        while (p_ != end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r'))
            ++p_;
    }
    bool Take(char c)
    {
        Space();
        if (p_ != end_ && *p_ == c)
        {
            ++p_;
            return true;
        }
        return false;
    }
    void Word(const char *w)
    {
        const size_t n = std::strlen(w);
        if (static_cast<size_t>(end_ - p_) < n || std::strncmp(p_, w, n) != 0)
            Fail("unknown literal");
        p_ += n;
    }
    static void Utf8(std::string &out, uint32_t c)
    {
        if (c < 0x80)
            out += static_cast<char>(c);
        else if (c < 0x800)
            out += static_cast<char>(0xC0 | (c >> 6)), out += static_cast<char>(0x80 | (c & 0x3F));
        else if (c < 0x10000)
            out += static_cast<char>(0xE0 | (c >> 12)), out += static_cast<char>(0x80 | ((c >> 6) & 0x3F)), out += static_cast<char>(0x80 | (c & 0x3F));
        else
            out += static_cast<char>(0xF0 | (c >> 18)), out += static_cast<char>(0x80 | ((c >> 12) & 0x3F)),
                out += static_cast<char>(0x80 | ((c >> 6) & 0x3F)), out += static_cast<char>(0x80 | (c & 0x3F));
    }
    uint32_t Hex4()
    {
        if (end_ - p_ < 4)
            Fail("short \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i, ++p_)
        {
            const char c = *p_;
            v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : (Fail("bad \\u escape"), 0));
        }
        return v;
    }
    std::string Text()
    {
        std::string out;
        for (;;)
        {
            if (p_ == end_)
                Fail("unterminated string");
            const char c = *p_++;
            if (c == '"')
                return out;
            if (c != '\\')
            {
                out += c;
                continue;
            }
            if (p_ == end_)
                Fail("unterminated escape");
            const char e = *p_++;
            switch (e)
            {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u':
            {
                uint32_t c0 = Hex4();
                if (c0 >= 0xD800 && c0 < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u')
                {
                    p_ += 2;
                    const uint32_t c1 = Hex4();
                    c0 = 0x10000 + ((c0 - 0xD800) << 10) + (c1 - 0xDC00);
                }
                Utf8(out, c0);
                break;
            }
            default: Fail("unknown escape");
            }
        }
    }
    Json Value(int depth)
    {
        if (depth > 256)
            Fail("nested too deeply");
        Space();
        if (p_ == end_)
            Fail("unexpected end");
        Json v;
        const char c = *p_;
        if (c == '{')
        {
            ++p_;
            v.kind = Json::kObject;
            if (Take('}'))
                return v;
            do
            {
                Space();
                if (p_ == end_ || *p_ != '"')
                    Fail("member name expected");
                ++p_;
                std::string key = Text();
                if (!Take(':'))
                    Fail("':' expected");
                v.object.emplace_back(std::move(key), Value(depth + 1));
            } while (Take(','));
            if (!Take('}'))
                Fail("'}' expected");
        }
        else if (c == '[')
        {
            ++p_;
            v.kind = Json::kArray;
            if (Take(']'))
                return v;
            do
                v.array.push_back(Value(depth + 1));
            while (Take(','));
            if (!Take(']'))
                Fail("']' expected");
        }
        else if (c == '"')
        {
            ++p_;
            v.kind = Json::kString, v.string = Text();
        }
        else if (c == 't')
            Word("true"), v.kind = Json::kBool, v.boolean = true;
        else if (c == 'f')
            Word("false"), v.kind = Json::kBool;
        else if (c == 'n')
            Word("null");
        else
        {
            const char *q = p_;
            while (q != end_ && (std::strchr("+-.eE", *q) != nullptr || (*q >= '0' && *q <= '9')))
                ++q;
            if (q == p_)
                Fail("value expected");
            const std::string text(p_, q);
            char *stop = nullptr;
            v.kind = Json::kNumber, v.number = std::strtod(text.c_str(), &stop);
            if (stop == text.c_str() || *stop != '\0')
                Fail("malformed number");
            p_ = q;
        }
        return v;
    }
    const char *p_, *end_;
    std::string what_;
};

std::vector<uint8_t> ReadFile(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw std::runtime_error("read file '" + path + "' failed.");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

std::vector<uint8_t> DecodeBase64(const std::string &text, size_t from, const std::string &what)
{
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = from; i < text.size(); ++i)
    {
        const char c = text[i];
        int v;
        if (c >= 'A' && c <= 'Z')
            v = c - 'A';
        else if (c >= 'a' && c <= 'z')
            v = c - 'a' + 26;
        else if (c >= '0' && c <= '9')
            v = c - '0' + 52;
        else if (c == '+' || c == '-')
            v = 62;
        else if (c == '/' || c == '_')
            v = 63;
        else if (c == '=' || c == '\n' || c == '\r')
            continue;
        else
            throw std::runtime_error("malformed base64 buffer in '" + what + "'.");
        acc = (acc << 6) | static_cast<uint32_t>(v), bits += 6;
        if (bits >= 8)
            bits -= 8, out.push_back(static_cast<uint8_t>((acc >> bits) & 0xFFu));
    }
    return out;
}

struct Document
{
    std::string path, directory;
    Json root;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<bool> loaded;
    std::vector<uint8_t> glb_chunk; // the binary chunk of a .glb: buffer 0 when that buffer has no uri
    bool has_glb_chunk = false;

    const Json &Entry(const char *table, long index) const
    {
        const Json *t = root.Find(table);
        if (!t || t->kind != Json::kArray || index < 0 || static_cast<size_t>(index) >= t->array.size())
            throw std::runtime_error(std::string("'") + path + "': " + table + " index out of range.");
        return t->array[static_cast<size_t>(index)];
    }
    const std::vector<uint8_t> &Buffer(long index)
    {
        const Json &b = Entry("buffers", index);
        const size_t k = static_cast<size_t>(index);
        if (buffers.size() <= k)
            buffers.resize(k + 1), loaded.resize(k + 1, false);
        if (!loaded[k])
        {
            const std::string uri = b.String("uri");
            if (uri.empty())
            {
                if (!has_glb_chunk || index != 0)
                    throw std::runtime_error("'" + path + "': buffer without uri outside a .glb.");
                buffers[k] = glb_chunk;
            }
            else if (uri.compare(0, 5, "data:") == 0)
            {
                const size_t comma = uri.find(',');
                if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma)
                    throw std::runtime_error("'" + path + "': only base64 data URIs are read.");
                buffers[k] = DecodeBase64(uri, comma + 1, path);
            }
            else
            {
                std::string name; // percent-decoding of the relative path
                for (size_t i = 0; i < uri.size(); ++i)
                    if (uri[i] == '%' && i + 2 < uri.size())
                        name += static_cast<char>(std::strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16)), i += 2;
                    else
                        name += uri[i];
                buffers[k] = ReadFile(directory + name);
            }
            if (static_cast<double>(buffers[k].size()) < b.Number("byteLength", 0.0))
                throw std::runtime_error("'" + path + "': buffer shorter than its byteLength.");
            loaded[k] = true;
        }
        return buffers[k];
    }
};

int ComponentBytes(long component_type)
{
    switch (component_type)
    {
    case 5120: case 5121: return 1; // BYTE, UNSIGNED_BYTE
    case 5122: case 5123: return 2; // SHORT, UNSIGNED_SHORT
    case 5125: case 5126: return 4; // UNSIGNED_INT, FLOAT
    default: return 0;
    }
}

int TypeComponents(const std::string &type)
{
    return type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
}

// A byte count / offset / element count from the (untrusted) JSON: a finite, non-negative number no larger than `limit`.
// (A double cast to size_t outside its range is undefined behaviour, and `count * stride` of two huge values wraps.)
size_t CheckedSize(const Document &doc, double v, size_t limit, const char *what)
{
    if (!(v >= 0.0) || !(v <= static_cast<double>(limit)) || v != std::floor(v))
        throw std::runtime_error("'" + doc.path + "': " + what + " out of range.");
    return static_cast<size_t>(v);
}

// Where an accessor's bytes are: [base, base + (count - 1) * stride + element) must lie inside its buffer view, and the view
// inside its buffer.  All comparisons are made without a product that could wrap.
struct AccessorSpan
{
    const uint8_t *data;
    size_t stride;
};
AccessorSpan SpanOf(Document &doc, const Json &a, const Json &view, size_t count, size_t element)
{
    const std::vector<uint8_t> &buffer = doc.Buffer(view.Index("buffer"));
    const size_t view_offset = CheckedSize(doc, view.Number("byteOffset", 0.0), buffer.size(), "bufferView.byteOffset");
    const size_t view_length = view.Find("byteLength") ? CheckedSize(doc, view.Number("byteLength", 0.0), buffer.size() - view_offset, "bufferView.byteLength")
                                                       : buffer.size() - view_offset;
    const size_t offset = CheckedSize(doc, a.Number("byteOffset", 0.0), view_length, "accessor.byteOffset");
    const size_t stride = view.Find("byteStride") && view.Number("byteStride", 0.0) != 0.0 ? CheckedSize(doc, view.Number("byteStride", 0.0), size_t(1) << 20, "bufferView.byteStride") : element;
    const size_t room = view_length - offset; // bytes from the first element to the end of the view
    if (element == 0 || stride == 0 || room < element || (count - 1) > (room - element) / stride)
        throw std::runtime_error("'" + doc.path + "': accessor reaches beyond its buffer.");
    return AccessorSpan{buffer.data() + view_offset + offset, stride};
}

// An accessor as `components` floats per element (integers converted, normalised ones scaled as the specification says).
// Sparse accessors are not read.
std::vector<float> ReadFloats(Document &doc, long accessor, int components, size_t *count_out)
{
    const Json &a = doc.Entry("accessors", accessor);
    const long ctype = a.Index("componentType"), view_index = a.Index("bufferView");
    const int cbytes = ComponentBytes(ctype), n = TypeComponents(a.String("type"));
    // (an element count beyond 2^32 cannot be indexed by a 32-bit face index anyway; it also keeps count * components small)
    const size_t count = CheckedSize(doc, a.Number("count", 0.0), size_t(1) << 32, "accessor.count");
    if (cbytes == 0 || n < components || a.Find("sparse") != nullptr)
        throw std::runtime_error("'" + doc.path + "': accessor of an unsupported kind.");
    *count_out = count;
    if (view_index < 0 || count == 0)
    {
        // an accessor without a buffer view is all zeros (specification) — of a size the FILE did not have to back with bytes: bounded
        // (round 5's advisor: `count` alone could ask for a 64 GB allocation), 2^26 elements are 1 GB of float4 already
        if (count > (size_t(1) << 26))
            throw std::runtime_error("'" + doc.path + "': an accessor without a buffer view names more than 2^26 elements.");
        return std::vector<float>(count * static_cast<size_t>(components), 0.0f);
    }
    const Json &view = doc.Entry("bufferViews", view_index);
    const size_t element = static_cast<size_t>(cbytes) * static_cast<size_t>(n);
    const AccessorSpan span = SpanOf(doc, a, view, count, element); // (throws before anything of `count` elements is allocated)
    const size_t stride = span.stride;
    std::vector<float> out(count * static_cast<size_t>(components), 0.0f);
    const bool normalized = a.Find("normalized") != nullptr && a.Find("normalized")->boolean;
    for (size_t i = 0; i < count; ++i)
        for (int c = 0; c < components; ++c)
        {
            const uint8_t *p = span.data + i * stride + static_cast<size_t>(c) * static_cast<size_t>(cbytes);
            float v = 0.0f;
            switch (ctype)
            {
            case 5126: std::memcpy(&v, p, 4); break;
            case 5120: { int8_t x; std::memcpy(&x, p, 1); v = normalized ? std::fmax(x / 127.0f, -1.0f) : float(x); break; }
            case 5121: { uint8_t x; std::memcpy(&x, p, 1); v = normalized ? x / 255.0f : float(x); break; }
            case 5122: { int16_t x; std::memcpy(&x, p, 2); v = normalized ? std::fmax(x / 32767.0f, -1.0f) : float(x); break; }
            case 5123: { uint16_t x; std::memcpy(&x, p, 2); v = normalized ? x / 65535.0f : float(x); break; }
            case 5125: { uint32_t x; std::memcpy(&x, p, 4); v = float(x); break; }
            }
            out[i * static_cast<size_t>(components) + static_cast<size_t>(c)] = v;
        }
    return out;
}

std::vector<uint32_t> ReadIndices(Document &doc, long accessor)
{
    const Json &a = doc.Entry("accessors", accessor);
    const long ctype = a.Index("componentType"), view_index = a.Index("bufferView");
    const size_t count = CheckedSize(doc, a.Number("count", 0.0), size_t(1) << 32, "accessor.count");
    const int cbytes = ComponentBytes(ctype);
    if ((ctype != 5121 && ctype != 5123 && ctype != 5125) || a.String("type") != "SCALAR" || view_index < 0 || a.Find("sparse") != nullptr)
        throw std::runtime_error("'" + doc.path + "': index accessor of an unsupported kind.");
    if (count == 0)
        return {};
    const Json &view = doc.Entry("bufferViews", view_index);
    const AccessorSpan span = SpanOf(doc, a, view, count, static_cast<size_t>(cbytes)); // (indices are tightly packed or strided alike)
    std::vector<uint32_t> out(count);
    for (size_t i = 0; i < count; ++i)
    {
        const uint8_t *p = span.data + i * span.stride;
        if (ctype == 5121)
            out[i] = *p;
        else if (ctype == 5123)
        {
            uint16_t x;
            std::memcpy(&x, p, 2), out[i] = x;
        }
        else
            std::memcpy(&out[i], p, 4);
    }
    return out;
}

// One imported mesh = one triangle primitive.
struct Imported
{
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> faces; // 3 per triangle, into this mesh's vertices
};

bool ImportPrimitive(Document &doc, const Json &primitive, Imported &out)
{
    const long mode = primitive.Find("mode") ? primitive.Index("mode") : 4;
    if (mode != 4 && mode != 5 && mode != 6)
        return false; // points and lines: nothing a triangle mesh holds (assimp sorts them into meshes the renderer never had faces for)
    const Json *attributes = primitive.Find("attributes");
    if (!attributes || attributes->Index("POSITION") < 0)
        return false;
    size_t n_vertices = 0, n = 0;
    out.positions = ReadFloats(doc, attributes->Index("POSITION"), 3, &n_vertices);
    if (attributes->Index("NORMAL") >= 0)
    {
        out.normals = ReadFloats(doc, attributes->Index("NORMAL"), 3, &n);
        if (n != n_vertices)
            out.normals.clear();
    }
    if (attributes->Index("TEXCOORD_0") >= 0)
    {
        out.texcoords = ReadFloats(doc, attributes->Index("TEXCOORD_0"), 2, &n);
        if (n != n_vertices)
            out.texcoords.clear();
        for (size_t i = 1; i < out.texcoords.size(); i += 2)
            out.texcoords[i] = 1.0f - out.texcoords[i]; // (assimp's glTF importer hands over 1 - v)
    }
    std::vector<uint32_t> order;
    if (primitive.Index("indices") >= 0)
        order = ReadIndices(doc, primitive.Index("indices"));
    else
    {
        order.resize(n_vertices);
        for (size_t i = 0; i < n_vertices; ++i)
            order[i] = static_cast<uint32_t>(i);
    }
    for (uint32_t i : order)
        if (i >= n_vertices)
            throw std::runtime_error("'" + doc.path + "': vertex index out of range.");
    if (mode == 4)
        out.faces.assign(order.begin(), order.begin() + static_cast<long>(order.size() - order.size() % 3));
    else
        for (size_t k = 2; k < order.size(); ++k)
        {
            if (mode == 5) // strip: every other triangle turned so that all wind alike
                out.faces.insert(out.faces.end(), {order[k - 2 + (k & 1u)], order[k - 1 - (k & 1u)], order[k]});
            else // fan
                out.faces.insert(out.faces.end(), {order[0], order[k - 1], order[k]});
        }
    return !out.faces.empty();
}

// model_loader.cpp:335-419, on the node tree of the file (transforms ignored there, and here).
MeshData FlattenNode(Document &doc, long node_index, uint32_t index_offset, bool &all_normals, bool &all_texcoords, int depth)
{
    if (depth > 256)
        throw std::runtime_error("'" + doc.path + "': node tree too deep (a cycle?).");
    const Json &node = doc.Entry("nodes", node_index);
    MeshData info;
    if (node.Index("mesh") >= 0)
    {
        const Json &mesh = doc.Entry("meshes", node.Index("mesh"));
        const Json *primitives = mesh.Find("primitives");
        if (primitives && primitives->kind == Json::kArray)
            for (const Json &primitive : primitives->array)
            {
                Imported m;
                if (!ImportPrimitive(doc, primitive, m))
                    continue;
                for (uint32_t i : m.faces)
                    info.indices.push_back(index_offset + i); // (the same offset for every mesh of the node: model_loader.cpp:345-347)
                info.texcoords.insert(info.texcoords.end(), m.texcoords.begin(), m.texcoords.end());
                info.positions.insert(info.positions.end(), m.positions.begin(), m.positions.end());
                info.normals.insert(info.normals.end(), m.normals.begin(), m.normals.end());
                all_normals = all_normals && !m.normals.empty();
                all_texcoords = all_texcoords && !m.texcoords.empty();
            }
    }
    const Json *children = node.Find("children");
    if (children && children->kind == Json::kArray)
        for (const Json &child : children->array)
        {
            if (child.kind != Json::kNumber)
                continue;
            // (offset = triangles gathered so far, not vertices: model_loader.cpp:395-396)
            const MeshData local = FlattenNode(doc, static_cast<long>(child.number), static_cast<uint32_t>(info.indices.size() / 3), all_normals, all_texcoords, depth + 1);
            info.indices.insert(info.indices.end(), local.indices.begin(), local.indices.end());
            info.texcoords.insert(info.texcoords.end(), local.texcoords.begin(), local.texcoords.end());
            info.positions.insert(info.positions.end(), local.positions.begin(), local.positions.end());
            info.normals.insert(info.normals.end(), local.normals.begin(), local.normals.end());
        }
    return info;
}

} // namespace

MeshData LoadGltf(const std::string &path, bool face_normals)
{
    Document doc;
    doc.path = path;
    const size_t slash = path.find_last_of("/\\");
    doc.directory = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    const std::vector<uint8_t> file = ReadFile(path);
    const char *json_begin = reinterpret_cast<const char *>(file.data()), *json_end = json_begin + file.size();
    if (file.size() >= 12 && std::memcmp(file.data(), "glTF", 4) == 0)
    {
        // .glb: 12-byte header, then chunks {length, type, data}: JSON first, an optional binary chunk second
        uint32_t version, total;
        std::memcpy(&version, file.data() + 4, 4), std::memcpy(&total, file.data() + 8, 4);
        if (version != 2 || total > file.size())
            throw std::runtime_error("'" + path + "': not a version-2 binary glTF.");
        size_t at = 12;
        bool have_json = false;
        while (at + 8 <= total)
        {
            uint32_t length, type;
            std::memcpy(&length, file.data() + at, 4), std::memcpy(&type, file.data() + at + 4, 4);
            at += 8;
            if (at + length > total)
                throw std::runtime_error("'" + path + "': chunk reaches beyond the file.");
            if (type == 0x4E4F534Au && !have_json) // "JSON"
                json_begin = reinterpret_cast<const char *>(file.data() + at), json_end = json_begin + length, have_json = true;
            else if (type == 0x004E4942u && !doc.has_glb_chunk) // "BIN\0"
                doc.glb_chunk.assign(file.begin() + static_cast<long>(at), file.begin() + static_cast<long>(at + length)), doc.has_glb_chunk = true;
            at += (length + 3u) & ~size_t(3);
        }
        if (!have_json)
            throw std::runtime_error("'" + path + "': binary glTF without a JSON chunk.");
    }
    doc.root = JsonReader(json_begin, json_end, path).Parse();
    if (doc.root.kind != Json::kObject)
        throw std::runtime_error("'" + path + "': not a glTF document.");

    // the node tree assimp builds: the default scene's roots under one root node (a single root IS the root node)
    std::vector<long> roots;
    const Json *scenes = doc.root.Find("scenes");
    if (scenes && scenes->kind == Json::kArray && !scenes->array.empty())
    {
        long which = doc.root.Index("scene");
        if (which < 0 || static_cast<size_t>(which) >= scenes->array.size())
            which = 0;
        const Json *nodes = scenes->array[static_cast<size_t>(which)].Find("nodes");
        if (nodes && nodes->kind == Json::kArray)
            for (const Json &n : nodes->array)
                if (n.kind == Json::kNumber)
                    roots.push_back(static_cast<long>(n.number));
    }
    MeshData m;
    bool all_normals = true, all_texcoords = true;
    for (long root : roots)
    {
        // (several roots hang under a synthetic root node without meshes: each is one of ITS children)
        const uint32_t offset = roots.size() == 1 ? 0u : static_cast<uint32_t>(m.indices.size() / 3);
        const MeshData local = FlattenNode(doc, root, offset, all_normals, all_texcoords, 0);
        m.indices.insert(m.indices.end(), local.indices.begin(), local.indices.end());
        m.texcoords.insert(m.texcoords.end(), local.texcoords.begin(), local.texcoords.end());
        m.positions.insert(m.positions.end(), local.positions.begin(), local.positions.end());
        m.normals.insert(m.normals.end(), local.normals.begin(), local.normals.end());
    }
    if (m.indices.empty())
        throw std::runtime_error("no triangles in '" + path + "'.");
    const size_t n_vertices = m.positions.size() / 3;
    for (uint32_t i : m.indices)
        if (i >= n_vertices)
            throw std::runtime_error("'" + path + "': the reference's flattening of several meshes (index offset = triangles gathered so far, "
                                     "model_loader.cpp:345-347, 395-396) leaves indices beyond the vertex array for this file; the "
                                     "reference would read outside its arrays here.");
    // per-vertex arrays only when every mesh brought them (a mixed file would leave arrays shorter than the vertices)
    if (!all_normals || m.normals.size() != m.positions.size())
        m.normals.clear();
    if (!all_texcoords || m.texcoords.size() / 2 != n_vertices)
        m.texcoords.clear();
    if (m.normals.empty() && !face_normals)
        GenerateSmoothNormals(m);
    const char *tangents = std::getenv("MCPT_MESH_TANGENTS"); // (asset_io.cpp: "uv" leaves the frames to the commit's uv rule)
    if (!(tangents && std::string(tangents) == "uv"))
        CalcTangentSpace(m);
    if (face_normals)
        m.normals.clear(); // not handed over (model_loader.cpp:363)
    return m;
}

} // namespace mcpt
