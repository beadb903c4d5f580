// Baseline JPEG reader of the scene front end.
//
// The reference reads JPEG textures through stb_image (extern/stb/stb_image.h, v2.28;
// src/utils/image_io.cpp:127-147).  This is an independent implementation of the JPEG
// baseline process (ITU T.81: Huffman coding, 8x8 DCT blocks, interleaved MCUs, restart
// intervals) that makes the same numerical choices as that library so that the texels
// agree: the jidctint-style integer inverse DCT (12-bit constants, +512 >> 10 after the
// column pass, +65536 + (128 << 17) >> 17 after the row pass), the 3:1 triangle filter
// for 2x chroma upsampling with the nearer chroma row weighted 3, and the 20-bit fixed
// point YCbCr conversion.  Not handled: progressive and arithmetic-coded files, 12-bit
// samples, CMYK.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "asset_io.hpp"

namespace mcpt
{
namespace
{

const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffmanTable
{
    // canonical code: for each length 1..16 the first code, the index of its first symbol and
    // the number of codes
    int first_code[17] = {}, first_index[17] = {}, count[17] = {};
    uint8_t symbols[256] = {};
    bool defined = false;
};

struct Component
{
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int rows = 0, cols = 0;   // meaningful samples
    int stride = 0, lines = 0; // padded to whole MCUs
    std::vector<uint8_t> samples;
};

class Decoder
{
public:
    Decoder(const std::vector<uint8_t> &file, const std::string &path) : f_(file), path_(path) {}

    void Run(int &width, int &height, int &channel, std::vector<uint8_t> &pixels)
    {
        if (f_.size() < 4 || f_[0] != 0xFF || f_[1] != 0xD8)
            Fail("not a JPEG file");
        at_ = 2;
        for (;;)
        {
            const int marker = NextMarker();
            if (marker == 0xD9)
                Fail("no image data");
            const size_t len = Be16();
            if (len < 2 || at_ + len - 2 > f_.size())
                Fail("truncated segment");
            const size_t end = at_ + len - 2;
            switch (marker)
            {
            case 0xC0:
            case 0xC1:
                ReadFrame();
                break;
            case 0xC2:
                Fail("progressive JPEG is not supported");
            case 0xC9:
            case 0xCA:
                Fail("arithmetic-coded JPEG is not supported");
            case 0xC4:
                ReadHuffmanTables(end);
                break;
            case 0xDB:
                ReadQuantTables(end);
                break;
            case 0xDD:
                restart_interval_ = Be16();
                break;
            case 0xDA:
                ReadScanHeader();
                at_ = end;
                DecodeScan();
                Finish(width, height, channel, pixels);
                return;
            default:
                break; // APPn, COM, ...
            }
            at_ = end;
        }
    }

private:
    [[noreturn]] void Fail(const char *what) const { throw std::runtime_error(std::string(what) + ": '" + path_ + "'."); }
    int U8() // one payload byte; a truncated or crafted segment fails instead of reading past the file
    {
        if (at_ >= f_.size())
            Fail("truncated JPEG");
        return f_[at_++];
    }
    size_t Be16()
    {
        if (at_ + 2 > f_.size())
            Fail("truncated JPEG");
        const size_t v = (size_t(f_[at_]) << 8) | f_[at_ + 1];
        at_ += 2;
        return v;
    }
    int NextMarker()
    {
        while (at_ + 1 < f_.size())
        {
            if (f_[at_] == 0xFF && f_[at_ + 1] != 0x00 && f_[at_ + 1] != 0xFF)
            {
                const int m = f_[at_ + 1];
                at_ += 2;
                return m;
            }
            ++at_;
        }
        Fail("truncated JPEG");
    }

    void ReadQuantTables(size_t end)
    {
        while (at_ < end)
        {
            const int pq_tq = U8();
            const int precision = pq_tq >> 4, id = pq_tq & 15;
            if (id > 3 || precision > 1 || at_ + (precision ? 128u : 64u) > end || end > f_.size())
                Fail("bad quantisation table");
            for (int k = 0; k < 64; ++k)
            {
                const int q = precision ? static_cast<int>(Be16()) : U8();
                quant_[id][kZigZag[k]] = static_cast<uint16_t>(q); // stored in natural order
            }
        }
    }
    void ReadHuffmanTables(size_t end)
    {
        while (at_ < end)
        {
            const int tc_th = U8();
            const int cls = tc_th >> 4, id = tc_th & 15;
            if (cls > 1 || id > 3 || at_ + 16 > end || end > f_.size())
                Fail("bad Huffman table");
            HuffmanTable &t = tables_[cls][id];
            int total = 0, code = 0;
            for (int l = 1; l <= 16; ++l)
            {
                t.count[l] = f_[at_++];
                t.first_index[l] = total;
                t.first_code[l] = code;
                total += t.count[l];
                code = (code + t.count[l]) << 1;
            }
            if (total > 256 || at_ + total > end)
                Fail("bad Huffman table");
            std::memcpy(t.symbols, &f_[at_], total);
            at_ += total;
            t.defined = true;
        }
    }
    void ReadFrame()
    {
        if (U8() != 8)
            Fail("only 8-bit JPEG is supported");
        height_ = static_cast<int>(Be16()), width_ = static_cast<int>(Be16());
        const int n = U8();
        if (width_ <= 0 || height_ <= 0 || (n != 1 && n != 3))
            Fail("unsupported JPEG frame");
        comps_.assign(n, Component());
        for (Component &c : comps_)
        {
            c.id = U8();
            const int hv = U8();
            c.h = hv >> 4, c.v = hv & 15;
            c.tq = U8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                Fail("bad JPEG component");
            h_max_ = std::max(h_max_, c.h), v_max_ = std::max(v_max_, c.v);
        }
        mcus_x_ = (width_ + 8 * h_max_ - 1) / (8 * h_max_), mcus_y_ = (height_ + 8 * v_max_ - 1) / (8 * v_max_);
        for (Component &c : comps_)
        {
            c.cols = (width_ * c.h + h_max_ - 1) / h_max_, c.rows = (height_ * c.v + v_max_ - 1) / v_max_;
            c.stride = mcus_x_ * c.h * 8, c.lines = mcus_y_ * c.v * 8;
            c.samples.assign(static_cast<size_t>(c.stride) * c.lines, 0);
        }
    }
    void ReadScanHeader()
    {
        const int n = U8();
        if (comps_.empty() || n != static_cast<int>(comps_.size()))
            Fail("unsupported JPEG scan (non-interleaved scans are not handled)");
        for (int i = 0; i < n; ++i)
        {
            const int id = U8(), tables = U8();
            Component *c = nullptr;
            for (Component &k : comps_)
                if (k.id == id)
                    c = &k;
            if (!c)
                Fail("bad JPEG scan");
            c->td = tables >> 4, c->ta = tables & 15;
            if (c->td > 3 || c->ta > 3)
                Fail("bad JPEG scan");
        }
    }

    // ---- entropy-coded data ----
    int ReadBit()
    {
        if (bit_count_ == 0)
        {
            int byte = 0;
            if (at_ < f_.size())
            {
                byte = f_[at_++];
                if (byte == 0xFF)
                {
                    const int next = at_ < f_.size() ? f_[at_] : 0xD9;
                    if (next == 0x00)
                        ++at_; // stuffed zero
                    else
                    {
                        --at_; // a marker: feed zeros from here on
                        byte = 0;
                        hit_marker_ = true;
                    }
                }
            }
            bit_buffer_ = byte, bit_count_ = 8;
        }
        --bit_count_;
        return (bit_buffer_ >> bit_count_) & 1;
    }
    int Receive(int n)
    {
        int v = 0;
        for (int i = 0; i < n; ++i)
            v = (v << 1) | ReadBit();
        return v;
    }
    static int Extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }
    int DecodeSymbol(const HuffmanTable &t)
    {
        int code = 0;
        for (int l = 1; l <= 16; ++l)
        {
            code = (code << 1) | ReadBit();
            if (t.count[l] && code - t.first_code[l] < t.count[l] && code >= t.first_code[l])
                return t.symbols[t.first_index[l] + code - t.first_code[l]];
        }
        Fail("corrupt JPEG data");
    }
    void DecodeBlock(Component &c, short block[64])
    {
        std::memset(block, 0, 64 * sizeof(short));
        const HuffmanTable &dc = tables_[0][c.td], &ac = tables_[1][c.ta];
        if (!dc.defined || !ac.defined)
            Fail("missing Huffman table");
        const int t = DecodeSymbol(dc);
        if (t > 16)
            Fail("corrupt JPEG data");
        c.dc_pred += Extend(Receive(t), t);
        block[0] = static_cast<short>(c.dc_pred * quant_[c.tq][0]);
        for (int k = 1; k < 64;)
        {
            const int rs = DecodeSymbol(ac), s = rs & 15, r = rs >> 4;
            if (s == 0)
            {
                if (rs != 0xF0)
                    break; // end of block
                k += 16;
                continue;
            }
            k += r;
            if (k > 63)
                Fail("corrupt JPEG data");
            const int zig = kZigZag[k++];
            block[zig] = static_cast<short>(Extend(Receive(s), s) * quant_[c.tq][zig]);
        }
    }

    // jidctint-style integer IDCT, same fixed point choices as stb_image (see the header)
    static int F2F(double x) { return static_cast<int>(x * 4096 + 0.5); }
    struct Butterfly
    {
        int x0, x1, x2, x3, t0, t1, t2, t3;
    };
    static Butterfly Idct1D(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
    {
        static const int c0541 = F2F(0.5411961f), cm1847 = F2F(-1.847759065f), c0765 = F2F(0.765366865f),
                         c1175 = F2F(1.175875602f), c0298 = F2F(0.298631336f), c2053 = F2F(2.053119869f),
                         c3072 = F2F(3.072711026f), c1501 = F2F(1.501321110f), cm0899 = F2F(-0.899976223f),
                         cm2562 = F2F(-2.562915447f), cm1961 = F2F(-1.961570560f), cm0390 = F2F(-0.390180644f);
        Butterfly b;
        int p2 = s2, p3 = s6;
        int p1 = (p2 + p3) * c0541;
        int t2 = p1 + p3 * cm1847, t3 = p1 + p2 * c0765;
        p2 = s0, p3 = s4;
        int t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
        b.x0 = t0 + t3, b.x3 = t0 - t3, b.x1 = t1 + t2, b.x2 = t1 - t2;
        t0 = s7, t1 = s5, t2 = s3, t3 = s1;
        p3 = t0 + t2;
        int p4 = t1 + t3;
        p1 = t0 + t3, p2 = t1 + t2;
        const int p5 = (p3 + p4) * c1175;
        t0 *= c0298, t1 *= c2053, t2 *= c3072, t3 *= c1501;
        p1 = p5 + p1 * cm0899, p2 = p5 + p2 * cm2562, p3 *= cm1961, p4 *= cm0390;
        b.t3 = t3 + p1 + p4, b.t2 = t2 + p2 + p3, b.t1 = t1 + p2 + p4, b.t0 = t0 + p1 + p3;
        return b;
    }
    static uint8_t Clamp(int x) { return static_cast<uint8_t>(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    static void Idct(const short d[64], uint8_t *out, int stride)
    {
        int v[64];
        for (int i = 0; i < 8; ++i)
        {
            if (!d[8 + i] && !d[16 + i] && !d[24 + i] && !d[32 + i] && !d[40 + i] && !d[48 + i] && !d[56 + i])
            {
                const int dc = d[i] * 4;
                for (int r = 0; r < 8; ++r)
                    v[8 * r + i] = dc;
                continue;
            }
            Butterfly b = Idct1D(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
            b.x0 += 512, b.x1 += 512, b.x2 += 512, b.x3 += 512;
            v[i] = (b.x0 + b.t3) >> 10, v[56 + i] = (b.x0 - b.t3) >> 10;
            v[8 + i] = (b.x1 + b.t2) >> 10, v[48 + i] = (b.x1 - b.t2) >> 10;
            v[16 + i] = (b.x2 + b.t1) >> 10, v[40 + i] = (b.x2 - b.t1) >> 10;
            v[24 + i] = (b.x3 + b.t0) >> 10, v[32 + i] = (b.x3 - b.t0) >> 10;
        }
        for (int r = 0; r < 8; ++r)
        {
            const int *w = v + 8 * r;
            Butterfly b = Idct1D(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
            const int bias = 65536 + (128 << 17);
            b.x0 += bias, b.x1 += bias, b.x2 += bias, b.x3 += bias;
            uint8_t *o = out + static_cast<size_t>(stride) * r;
            o[0] = Clamp((b.x0 + b.t3) >> 17), o[7] = Clamp((b.x0 - b.t3) >> 17);
            o[1] = Clamp((b.x1 + b.t2) >> 17), o[6] = Clamp((b.x1 - b.t2) >> 17);
            o[2] = Clamp((b.x2 + b.t1) >> 17), o[5] = Clamp((b.x2 - b.t1) >> 17);
            o[3] = Clamp((b.x3 + b.t0) >> 17), o[4] = Clamp((b.x3 - b.t0) >> 17);
        }
    }

    void DecodeScan()
    {
        int until_restart = restart_interval_;
        short block[64];
        for (int my = 0; my < mcus_y_; ++my)
            for (int mx = 0; mx < mcus_x_; ++mx)
            {
                for (Component &c : comps_)
                    for (int by = 0; by < c.v; ++by)
                        for (int bx = 0; bx < c.h; ++bx)
                        {
                            DecodeBlock(c, block);
                            uint8_t *dst = &c.samples[(static_cast<size_t>(my) * c.v + by) * 8 * c.stride +
                                                      (static_cast<size_t>(mx) * c.h + bx) * 8];
                            Idct(block, dst, c.stride);
                        }
                if (restart_interval_ && --until_restart == 0 && !(my == mcus_y_ - 1 && mx == mcus_x_ - 1))
                {
                    // align to the restart marker and reset the predictors
                    bit_count_ = 0, hit_marker_ = false;
                    const int m = NextMarker();
                    if (m < 0xD0 || m > 0xD7)
                        Fail("missing JPEG restart marker");
                    for (Component &c : comps_)
                        c.dc_pred = 0;
                    until_restart = restart_interval_;
                }
            }
    }

    // one output row of one component at full horizontal resolution
    void UpsampleRow(const Component &c, int y, std::vector<uint8_t> &out) const
    {
        const int hs = h_max_ / c.h, vs = v_max_ / c.v, w = c.cols;
        auto row = [&](int r) { return &c.samples[static_cast<size_t>(std::min(std::max(r, 0), c.rows - 1)) * c.stride]; };
        const uint8_t *near_row, *far_row;
        if (vs == 1)
            near_row = far_row = row(y);
        else if (vs == 2)
        {
            // output rows 2k+1 and 2k+2 lie between chroma rows k and k+1; the closer one is "near"
            const int k = (y + 1) / 2 - ((y & 1) ? 1 : 0);
            near_row = (y & 1) ? row(k) : row(y / 2);
            far_row = (y & 1) ? row(k + 1) : row(y / 2 - 1);
            if (y == 0)
                near_row = far_row = row(0);
        }
        else
            near_row = far_row = row(y / vs);
        out.resize(static_cast<size_t>(w) * hs + 8);
        if (hs == 1 && vs == 1)
            std::memcpy(out.data(), near_row, w);
        else if (hs == 1 && vs == 2)
            for (int i = 0; i < w; ++i)
                out[i] = static_cast<uint8_t>((3 * near_row[i] + far_row[i] + 2) >> 2);
        else if (hs == 2 && vs == 1)
        {
            const uint8_t *in = near_row;
            if (w == 1)
                out[0] = out[1] = in[0];
            else
            {
                out[0] = in[0];
                out[1] = static_cast<uint8_t>((in[0] * 3 + in[1] + 2) >> 2);
                int i = 1;
                for (; i < w - 1; ++i)
                {
                    const int n = 3 * in[i] + 2;
                    out[2 * i] = static_cast<uint8_t>((n + in[i - 1]) >> 2);
                    out[2 * i + 1] = static_cast<uint8_t>((n + in[i + 1]) >> 2);
                }
                out[2 * i] = static_cast<uint8_t>((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
                out[2 * i + 1] = in[w - 1];
            }
        }
        else if (hs == 2 && vs == 2)
        {
            if (w == 1)
                out[0] = out[1] = static_cast<uint8_t>((3 * near_row[0] + far_row[0] + 2) >> 2);
            else
            {
                int t1 = 3 * near_row[0] + far_row[0];
                out[0] = static_cast<uint8_t>((t1 + 2) >> 2);
                for (int i = 1; i < w; ++i)
                {
                    const int t0 = t1;
                    t1 = 3 * near_row[i] + far_row[i];
                    out[2 * i - 1] = static_cast<uint8_t>((3 * t0 + t1 + 8) >> 4);
                    out[2 * i] = static_cast<uint8_t>((3 * t1 + t0 + 8) >> 4);
                }
                out[2 * w - 1] = static_cast<uint8_t>((t1 + 2) >> 2);
            }
        }
        else // other ratios: nearest neighbour
            for (int i = 0; i < w; ++i)
                for (int j = 0; j < hs; ++j)
                    out[static_cast<size_t>(i) * hs + j] = near_row[i];
    }

    void Finish(int &width, int &height, int &channel, std::vector<uint8_t> &pixels) const
    {
        width = width_, height = height_, channel = static_cast<int>(comps_.size());
        pixels.resize(static_cast<size_t>(width_) * height_ * channel);
        std::vector<uint8_t> line[3];
        auto fixed = [](float x) { return static_cast<int>(x * 4096.0f + 0.5f) << 8; };
        const int cr_r = fixed(1.40200f), cr_g = -fixed(0.71414f), cb_g = -fixed(0.34414f), cb_b = fixed(1.77200f);
        for (int y = 0; y < height_; ++y)
        {
            for (size_t k = 0; k < comps_.size(); ++k)
                UpsampleRow(comps_[k], y, line[k]);
            uint8_t *out = &pixels[static_cast<size_t>(y) * width_ * channel];
            if (channel == 1)
            {
                std::memcpy(out, line[0].data(), width_);
                continue;
            }
            for (int x = 0; x < width_; ++x)
            {
                const int y_fixed = (line[0][x] << 20) + (1 << 19);
                const int cb = line[1][x] - 128, cr = line[2][x] - 128;
                const int r = (y_fixed + cr * cr_r) >> 20;
                const int g = (y_fixed + cr * cr_g + static_cast<int>((cb * cb_g) & 0xFFFF0000)) >> 20;
                const int b = (y_fixed + cb * cb_b) >> 20;
                out[3 * x] = Clamp(r), out[3 * x + 1] = Clamp(g), out[3 * x + 2] = Clamp(b);
            }
        }
    }

    const std::vector<uint8_t> &f_;
    std::string path_;
    size_t at_ = 0;
    int width_ = 0, height_ = 0, h_max_ = 1, v_max_ = 1, mcus_x_ = 0, mcus_y_ = 0, restart_interval_ = 0;
    std::vector<Component> comps_;
    uint16_t quant_[4][64] = {};
    HuffmanTable tables_[2][4];
    int bit_buffer_ = 0, bit_count_ = 0;
    bool hit_marker_ = false;
};

} // namespace

// 8-bit samples, `channel` per pixel (1 or 3), row 0 = top.
void LoadJpeg8(const std::string &path, int &width, int &height, int &channel, std::vector<uint8_t> &pixels)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("[error] load image '" + path + "' failed.");
    const std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    Decoder(file, path).Run(width, height, channel, pixels);
}

} // namespace mcpt
