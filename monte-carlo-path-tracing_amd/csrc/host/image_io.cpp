// Image writers.  PNG follows the reference's output conversion
// (src/utils/image_io.cpp:25-53: linear -> sRGB curve, scale by 255, truncate);
// EXR / PFM / raw keep the linear float32 frame.  zlib (already needed for the
// .serialized meshes) does the deflate of the PNG stream and of the EXR ZIP blocks.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <zlib.h>

#include "frontend.hpp"

namespace mcpt
{
namespace
{

std::string Suffix(const std::string &path)
{
    const size_t dot = path.find_last_of('.');
    return dot == std::string::npos ? "" : path.substr(dot + 1);
}

void WriteAll(const std::string &path, const std::vector<uint8_t> &bytes)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f)
        throw std::runtime_error("cannot write '" + path + "'.");
    const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    if (n != bytes.size())
        throw std::runtime_error("short write to '" + path + "'.");
}

struct Bytes
{
    std::vector<uint8_t> v;
    void u8(uint8_t x) { v.push_back(x); }
    void be32(uint32_t x)
    {
        for (int s = 24; s >= 0; s -= 8)
            v.push_back(static_cast<uint8_t>(x >> s));
    }
    void le16(uint16_t x) { v.push_back(x & 0xff), v.push_back(x >> 8); }
    void le32(uint32_t x)
    {
        for (int s = 0; s < 32; s += 8)
            v.push_back(static_cast<uint8_t>(x >> s));
    }
    void le64(uint64_t x)
    {
        for (int s = 0; s < 64; s += 8)
            v.push_back(static_cast<uint8_t>(x >> s));
    }
    void raw(const void *p, size_t n)
    {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        v.insert(v.end(), b, b + n);
    }
    void str(const char *s) { raw(s, std::strlen(s) + 1); }
};

uint32_t Crc32(const uint8_t *p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready)
    {
        for (uint32_t i = 0; i < 256; ++i)
        {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k)
                c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i)
        crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

void PngChunk(Bytes &out, const char *type, const std::vector<uint8_t> &data)
{
    out.be32(static_cast<uint32_t>(data.size()));
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.raw(body.data(), body.size());
    out.be32(Crc32(body.data(), body.size()));
}

// image_io.cpp:25-53
uint8_t ToSrgb8(float linear)
{
    float v = linear <= 0.0031308f ? 12.92f * linear : 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    return static_cast<uint8_t>(v * 255.0f);
}

void WritePng(const std::string &path, const float *frame, int w, int h)
{
    std::vector<uint8_t> scan;
    scan.reserve(static_cast<size_t>(h) * (1 + 3 * w));
    for (int y = 0; y < h; ++y)
    {
        scan.push_back(0); // filter: none
        for (int x = 0; x < 3 * w; ++x)
            scan.push_back(ToSrgb8(frame[static_cast<size_t>(y) * 3 * w + x]));
    }
    uLongf z_size = compressBound(static_cast<uLong>(scan.size()));
    std::vector<uint8_t> z(z_size);
    if (compress2(z.data(), &z_size, scan.data(), static_cast<uLong>(scan.size()), 6) != Z_OK)
        throw std::runtime_error("deflate failed for '" + path + "'.");
    z.resize(z_size);

    Bytes out;
    const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.raw(magic, 8);
    Bytes ihdr;
    ihdr.be32(w), ihdr.be32(h);
    ihdr.u8(8), ihdr.u8(2), ihdr.u8(0), ihdr.u8(0), ihdr.u8(0);
    PngChunk(out, "IHDR", ihdr.v);
    PngChunk(out, "IDAT", z);
    PngChunk(out, "IEND", {});
    WriteAll(path, out.v);
}

// OpenEXR 2.0 single-part scanline file, channels B, G, R as FLOAT, ZIP compression
// (blocks of 16 scanlines: bytes split into even / odd halves, delta-predicted,
// deflated; a block that does not shrink is stored raw, as the format prescribes),
// increasing-y line order.
void WriteExr(const std::string &path, const float *frame, int w, int h)
{
    Bytes out;
    out.le32(20000630u); // magic
    out.le32(2u);        // version 2, no flags
    auto attr = [&](const char *name, const char *type, const Bytes &value)
    {
        out.str(name), out.str(type);
        out.le32(static_cast<uint32_t>(value.v.size()));
        out.raw(value.v.data(), value.v.size());
    };
    Bytes channels;
    for (const char *c : {"B", "G", "R"})
    {
        channels.str(c);
        channels.le32(2); // FLOAT
        channels.u8(0), channels.u8(0), channels.u8(0), channels.u8(0);
        channels.le32(1), channels.le32(1);
    }
    channels.u8(0);
    attr("channels", "chlist", channels);
    Bytes comp;
    comp.u8(3); // ZIP_COMPRESSION
    attr("compression", "compression", comp);
    Bytes window;
    window.le32(0), window.le32(0), window.le32(w - 1), window.le32(h - 1);
    attr("dataWindow", "box2i", window);
    attr("displayWindow", "box2i", window);
    Bytes order;
    order.u8(0);
    attr("lineOrder", "lineOrder", order);
    Bytes one;
    const float f1 = 1.0f, f0 = 0.0f;
    one.raw(&f1, 4);
    attr("pixelAspectRatio", "float", one);
    Bytes centre;
    centre.raw(&f0, 4), centre.raw(&f0, 4);
    attr("screenWindowCenter", "v2f", centre);
    attr("screenWindowWidth", "float", one);
    out.u8(0); // end of header

    constexpr int kLinesPerBlock = 16;
    const int n_blocks = (h + kLinesPerBlock - 1) / kLinesPerBlock;
    std::vector<std::vector<uint8_t>> blocks(n_blocks);
    std::vector<float> plane(w);
    for (int k = 0; k < n_blocks; ++k)
    {
        const int y0 = k * kLinesPerBlock, y1 = std::min(h, y0 + kLinesPerBlock);
        Bytes raw;
        for (int y = y0; y < y1; ++y)
            for (int c = 2; c >= 0; --c) // B, G, R planes
            {
                for (int x = 0; x < w; ++x)
                    plane[x] = frame[(static_cast<size_t>(y) * w + x) * 3 + c];
                raw.raw(plane.data(), static_cast<size_t>(w) * 4);
            }
        const size_t n = raw.v.size();
        std::vector<uint8_t> shuffled(n);
        const size_t half = (n + 1) / 2;
        for (size_t i = 0; i < n; ++i)
            shuffled[(i & 1) ? half + i / 2 : i / 2] = raw.v[i];
        for (size_t i = n; i-- > 1;)
            shuffled[i] = static_cast<uint8_t>(shuffled[i] - shuffled[i - 1] + 128);
        uLongf z_size = compressBound(static_cast<uLong>(n));
        std::vector<uint8_t> z(z_size);
        if (compress2(z.data(), &z_size, shuffled.data(), static_cast<uLong>(n), 6) != Z_OK)
            throw std::runtime_error("deflate failed for '" + path + "'.");
        z.resize(z_size);
        blocks[k] = z.size() < n ? std::move(z) : std::move(raw.v);
    }
    uint64_t offset = out.v.size() + 8ull * n_blocks;
    for (int k = 0; k < n_blocks; ++k)
    {
        out.le64(offset);
        offset += 8 + blocks[k].size();
    }
    for (int k = 0; k < n_blocks; ++k)
    {
        out.le32(k * kLinesPerBlock);
        out.le32(static_cast<uint32_t>(blocks[k].size()));
        out.raw(blocks[k].data(), blocks[k].size());
    }
    WriteAll(path, out.v);
}

void WritePfm(const std::string &path, const float *frame, int w, int h)
{
    Bytes out;
    char header[64];
    const int n = std::snprintf(header, sizeof(header), "PF\n%d %d\n-1.0\n", w, h);
    out.raw(header, n);
    for (int y = h - 1; y >= 0; --y) // PFM stores the bottom row first
        out.raw(frame + static_cast<size_t>(y) * w * 3, static_cast<size_t>(w) * 12);
    WriteAll(path, out.v);
}

} // namespace

void WriteImage(const std::string &path, const float *frame, int width, int height)
{
    const std::string suffix = Suffix(path);
    if (suffix == "png")
        WritePng(path, frame, width, height);
    else if (suffix == "exr")
        WriteExr(path, frame, width, height);
    else if (suffix == "pfm")
        WritePfm(path, frame, width, height);
    else if (suffix == "f32" || suffix == "raw")
    {
        std::vector<uint8_t> bytes(static_cast<size_t>(width) * height * 12);
        std::memcpy(bytes.data(), frame, bytes.size());
        WriteAll(path, bytes);
    }
    else
        throw std::runtime_error("unsupported image suffix '." + suffix + "'.");
}

} // namespace mcpt
