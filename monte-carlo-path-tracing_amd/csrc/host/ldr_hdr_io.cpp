// PNG and Radiance .hdr readers of the scene front end.
//
// The reference reads both through stb_image (src/utils/image_io.cpp:99-147): 8-bit
// images become floats value/255 in the file's own channel count, .hdr files the RGBE
// floats.  Written from the format definitions (PNG: RFC 2083 chunks, zlib stream,
// five scanline filters; RGBE: Ward's run-length scanlines).  Not handled: Adam7
// interlacing, 2-channel (grey + alpha) PNGs (the renderer has no 2-channel bitmap).
#include <cmath>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include <zlib.h>

#include "asset_io.hpp"

namespace mcpt
{
namespace
{

std::vector<uint8_t> ReadFile(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw std::runtime_error("[error] load image '" + path + "' failed.");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

uint32_t Be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int Paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

} // namespace

// 8-bit samples, `channel` per pixel (1, 3 or 4), row 0 = top.
void LoadPng8(const std::string &path, int &width, int &height, int &channel, std::vector<uint8_t> &pixels)
{
    const std::vector<uint8_t> f = ReadFile(path);
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || std::memcmp(f.data(), magic, 8) != 0)
        throw std::runtime_error("not a PNG file: '" + path + "'.");
    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    for (size_t at = 8; at + 12 <= f.size();)
    {
        const uint32_t len = Be32(&f[at]);
        const std::string type(reinterpret_cast<const char *>(&f[at + 4]), 4);
        const uint8_t *data = &f[at + 8];
        if (at + 12 + len > f.size())
            throw std::runtime_error("truncated PNG '" + path + "'.");
        if (type == "IHDR" && len >= 13)
            w = Be32(data), h = Be32(data + 4), depth = data[8], color = data[9], interlace = data[12];
        else if (type == "PLTE")
            palette.assign(data, data + len);
        else if (type == "tRNS")
            trns.assign(data, data + len);
        else if (type == "IDAT")
            idat.insert(idat.end(), data, data + len);
        else if (type == "IEND")
            break;
        at += 12 + len;
    }
    if (!w || !h || idat.empty())
        throw std::runtime_error("malformed PNG '" + path + "'.");
    if (interlace)
        throw std::runtime_error("interlaced PNG is not supported: '" + path + "'.");
    const int samples = color == 0 ? 1 : (color == 2 ? 3 : (color == 3 ? 1 : (color == 4 ? 2 : (color == 6 ? 4 : 0))));
    if (!samples || (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16))
        throw std::runtime_error("unsupported PNG colour type in '" + path + "'.");
    const size_t bits_per_pixel = size_t(samples) * depth, stride = (bits_per_pixel * w + 7) / 8;
    const size_t bpp = std::max<size_t>(1, bits_per_pixel / 8); // filter distance in bytes
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &raw_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || raw_len != raw.size())
        throw std::runtime_error("corrupt PNG data in '" + path + "'.");
    std::vector<uint8_t> rows(stride * h);
    for (uint32_t y = 0; y < h; ++y)
    {
        const uint8_t filter = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *dst = &rows[stride * y];
        const uint8_t *up = y ? dst - stride : nullptr;
        for (size_t i = 0; i < stride; ++i)
        {
            const int a = i >= bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = src[i];
            switch (filter)
            {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += Paeth(a, b, c); break;
            default: throw std::runtime_error("corrupt PNG filter in '" + path + "'.");
            }
            dst[i] = static_cast<uint8_t>(v);
        }
    }
    // to 8-bit samples in the channel count stb_image reports for the file
    const bool palette_alpha = color == 3 && !trns.empty();
    channel = color == 3 ? (palette_alpha ? 4 : 3) : samples;
    if ((color == 0 || color == 2) && !trns.empty())
        throw std::runtime_error("PNG colour-key transparency is not supported: '" + path + "'.");
    if (channel == 2)
        throw std::runtime_error("grey + alpha PNG is not supported: '" + path + "'.");
    width = static_cast<int>(w), height = static_cast<int>(h);
    pixels.resize(size_t(w) * h * channel);
    for (uint32_t y = 0; y < h; ++y)
    {
        const uint8_t *row = &rows[stride * y];
        for (uint32_t x = 0; x < w; ++x)
        {
            uint8_t *out = &pixels[(size_t(y) * w + x) * channel];
            if (color == 3)
            {
                const size_t bit = size_t(x) * depth;
                const uint32_t index = (row[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1);
                if (3 * index + 2 >= palette.size())
                    throw std::runtime_error("PNG palette index out of range in '" + path + "'.");
                out[0] = palette[3 * index], out[1] = palette[3 * index + 1], out[2] = palette[3 * index + 2];
                if (palette_alpha)
                    out[3] = index < trns.size() ? trns[index] : 255;
            }
            else if (depth == 16)
            {
                for (int s = 0; s < samples; ++s)
                    out[s] = row[(size_t(x) * samples + s) * 2]; // high byte, as stb_image
            }
            else if (depth == 8)
            {
                for (int s = 0; s < samples; ++s)
                    out[s] = row[size_t(x) * samples + s];
            }
            else // 1, 2, 4-bit grey: scaled to 0..255
            {
                const size_t bit = size_t(x) * depth;
                const uint32_t v = (row[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1);
                out[0] = static_cast<uint8_t>(v * 255u / ((1u << depth) - 1));
            }
        }
    }
}

// Radiance RGBE -> 3 floats per pixel, row 0 = top (stb_image's stbi_loadf on .hdr).
ImageData LoadRadianceHdr(const std::string &path)
{
    const std::vector<uint8_t> f = ReadFile(path);
    size_t at = 0;
    auto line = [&]()
    {
        std::string s;
        while (at < f.size() && f[at] != '\n')
            s += static_cast<char>(f[at++]);
        ++at;
        return s;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE")
        throw std::runtime_error("not a Radiance HDR file: '" + path + "'.");
    bool rgbe = false;
    for (;;)
    {
        const std::string s = line();
        if (s.empty())
            break;
        if (s == "FORMAT=32-bit_rle_rgbe")
            rgbe = true;
        if (at >= f.size())
            break;
    }
    int w = 0, h = 0;
    if (!rgbe || std::sscanf(line().c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0)
        throw std::runtime_error("unsupported Radiance HDR layout in '" + path + "'.");
    ImageData img;
    img.width = w, img.height = h, img.channel = 3;
    img.data.resize(size_t(w) * h * 3);
    std::vector<uint8_t> scan(size_t(w) * 4);
    auto need = [&](size_t n)
    {
        if (at + n > f.size())
            throw std::runtime_error("truncated Radiance HDR '" + path + "'.");
    };
    for (int y = 0; y < h; ++y)
    {
        need(4);
        if (w >= 8 && w < 32768 && f[at] == 2 && f[at + 1] == 2 && !(f[at + 2] & 0x80))
        {
            if (((int(f[at + 2]) << 8) | f[at + 3]) != w)
                throw std::runtime_error("corrupt Radiance HDR scanline in '" + path + "'.");
            at += 4;
            for (int c = 0; c < 4; ++c) // each of R, G, B, E run-length coded on its own
                for (int x = 0; x < w;)
                {
                    need(1);
                    int count = f[at++];
                    if (count > 128)
                    {
                        count -= 128;
                        need(1);
                        const uint8_t v = f[at++];
                        if (x + count > w)
                            throw std::runtime_error("corrupt Radiance HDR run in '" + path + "'.");
                        for (int k = 0; k < count; ++k)
                            scan[size_t(x++) * 4 + c] = v;
                    }
                    else
                    {
                        need(count);
                        if (count == 0 || x + count > w)
                            throw std::runtime_error("corrupt Radiance HDR run in '" + path + "'.");
                        for (int k = 0; k < count; ++k)
                            scan[size_t(x++) * 4 + c] = f[at++];
                    }
                }
        }
        else
        {
            need(size_t(w) * 4); // flat RGBE pixels
            std::memcpy(scan.data(), &f[at], size_t(w) * 4);
            at += size_t(w) * 4;
        }
        for (int x = 0; x < w; ++x)
        {
            const uint8_t *p = &scan[size_t(x) * 4];
            float *out = &img.data[(size_t(y) * w + x) * 3];
            if (p[3] == 0)
                out[0] = out[1] = out[2] = 0.0f;
            else
            {
                const float scale = static_cast<float>(std::ldexp(1.0, int(p[3]) - (128 + 8)));
                out[0] = p[0] * scale, out[1] = p[1] * scale, out[2] = p[2] * scale;
            }
        }
    }
    return img;
}

} // namespace mcpt
