// PIZ block decoder for the OpenEXR reader in asset_io.cpp.
//
// The reference reads its environment maps with tinyexr (src/utils/image_io.cpp:
// 75-98); the matpreview scenes ship a PIZ-compressed envmap.exr.  Written from
// the OpenEXR file-format description of the PIZ scheme: a block is
//   u16 min_nonzero, u16 max_nonzero, bitmap[min..max] of the 16-bit values in
//   use, i32 length, canonical-Huffman coded u16 stream
// and decoding runs Huffman -> 2-D Haar-like wavelet per channel plane -> value
// look-up table -> per-scanline channel interleave.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "asset_io.hpp"

namespace mcpt
{
namespace
{

constexpr int kEncodeBits = 16, kDecodeBits = 14;
constexpr int kEncodeSize = (1 << kEncodeBits) + 1; // + the run-length symbol
constexpr int kDecodeSize = 1 << kDecodeBits;
constexpr int kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;

[[noreturn]] void Corrupt() { throw std::runtime_error("corrupt PIZ block in EXR file."); }

struct BitReader
{
    const uint8_t *p, *end;
    uint64_t acc = 0;
    int have = 0;
    uint32_t Take(int n)
    {
        while (have < n)
        {
            if (p >= end)
                Corrupt();
            acc = (acc << 8) | *p++;
            have += 8;
        }
        have -= n;
        return static_cast<uint32_t>((acc >> have) & ((1ull << n) - 1));
    }
};

// code word table entry: length in the low 6 bits, code above
inline int LengthOf(uint64_t e) { return static_cast<int>(e & 63); }
inline uint64_t CodeOf(uint64_t e) { return e >> 6; }

// packed code lengths (6 bits each, with zero-run escapes) -> canonical codes
const uint8_t *UnpackCodeTable(const uint8_t *p, const uint8_t *end, uint32_t first, uint32_t last,
                               std::vector<uint64_t> &table)
{
    table.assign(kEncodeSize, 0);
    BitReader bits{p, end};
    for (uint32_t s = first; s <= last; ++s)
    {
        const uint32_t len = bits.Take(6);
        table[s] = len;
        int run = 0;
        if (len == static_cast<uint32_t>(kLongZeroRun))
            run = static_cast<int>(bits.Take(8)) + kShortestLongRun;
        else if (len >= static_cast<uint32_t>(kShortZeroRun))
            run = static_cast<int>(len) - kShortZeroRun + 2;
        if (run)
        {
            if (s + run > last + 1)
                Corrupt();
            for (int i = 0; i < run; ++i)
                table[s + i] = 0;
            s += run - 1;
        }
    }
    // canonical assignment: shorter codes get numerically larger prefixes
    uint64_t next[59] = {};
    for (int i = 0; i < kEncodeSize; ++i)
        next[table[i]] += 1;
    uint64_t c = 0;
    for (int l = 58; l > 0; --l)
    {
        const uint64_t nc = (c + next[l]) >> 1;
        next[l] = c;
        c = nc;
    }
    for (int i = 0; i < kEncodeSize; ++i)
    {
        const int l = static_cast<int>(table[i]);
        if (l > 0)
            table[i] = static_cast<uint64_t>(l) | (next[l]++ << 6);
    }
    return bits.p;
}

struct DecodeEntry
{
    int len = 0;               // > 0: short code, `symbol` is the value
    uint32_t symbol = 0;
    std::vector<uint32_t> longs; // symbols whose codes share this 14-bit prefix
};

void BuildDecodeTable(const std::vector<uint64_t> &codes, uint32_t first, uint32_t last, std::vector<DecodeEntry> &dec)
{
    dec.assign(kDecodeSize, DecodeEntry());
    for (uint32_t s = first; s <= last; ++s)
    {
        const uint64_t c = CodeOf(codes[s]);
        const int l = LengthOf(codes[s]);
        if (c >> l)
            Corrupt();
        if (l > kDecodeBits)
        {
            DecodeEntry &e = dec[c >> (l - kDecodeBits)];
            if (e.len)
                Corrupt();
            e.longs.push_back(s);
        }
        else if (l)
        {
            const size_t base = static_cast<size_t>(c << (kDecodeBits - l));
            for (size_t i = 0; i < (size_t(1) << (kDecodeBits - l)); ++i)
            {
                dec[base + i].len = l;
                dec[base + i].symbol = s;
            }
        }
    }
}

void HuffmanDecode(const uint8_t *src, size_t n_src, uint16_t *out, size_t n_out)
{
    if (n_src < 20)
        Corrupt();
    uint32_t first, last, n_bits;
    std::memcpy(&first, src, 4);
    std::memcpy(&last, src + 4, 4);
    std::memcpy(&n_bits, src + 12, 4);
    if (first >= static_cast<uint32_t>(kEncodeSize) || last >= static_cast<uint32_t>(kEncodeSize))
        Corrupt();
    const uint8_t *end = src + n_src;
    std::vector<uint64_t> codes;
    const uint8_t *p = UnpackCodeTable(src + 20, end, first, last, codes);
    if (n_bits > 8ull * static_cast<size_t>(end - p))
        Corrupt();
    std::vector<DecodeEntry> dec;
    BuildDecodeTable(codes, first, last, dec);

    const uint32_t run_symbol = last;
    const uint8_t *stop = p + (n_bits + 7) / 8;
    uint64_t acc = 0;
    int have = 0;
    size_t produced = 0;
    auto emit = [&](uint32_t symbol)
    {
        if (symbol == run_symbol)
        {
            if (have < 8)
            {
                if (p >= end)
                    Corrupt();
                acc = (acc << 8) | *p++;
                have += 8;
            }
            have -= 8;
            const size_t count = (acc >> have) & 0xFF;
            if (produced == 0 || produced + count > n_out)
                Corrupt();
            const uint16_t v = out[produced - 1];
            for (size_t i = 0; i < count; ++i)
                out[produced++] = v;
        }
        else
        {
            if (produced >= n_out)
                Corrupt();
            out[produced++] = static_cast<uint16_t>(symbol);
        }
    };
    while (p < stop)
    {
        acc = (acc << 8) | *p++;
        have += 8;
        while (have >= kDecodeBits)
        {
            const DecodeEntry &e = dec[(acc >> (have - kDecodeBits)) & (kDecodeSize - 1)];
            if (e.len)
            {
                have -= e.len;
                emit(e.symbol);
                continue;
            }
            bool matched = false;
            for (uint32_t s : e.longs)
            {
                const int l = LengthOf(codes[s]);
                while (have < l && p < stop)
                {
                    acc = (acc << 8) | *p++;
                    have += 8;
                }
                if (have >= l && CodeOf(codes[s]) == ((acc >> (have - l)) & ((1ull << l) - 1)))
                {
                    have -= l;
                    emit(s);
                    matched = true;
                    break;
                }
            }
            if (!matched)
                Corrupt();
        }
    }
    const int pad = (8 - static_cast<int>(n_bits & 7)) & 7;
    acc >>= pad;
    have -= pad;
    while (have > 0)
    {
        const DecodeEntry &e = dec[(acc << (kDecodeBits - have)) & (kDecodeSize - 1)];
        if (!e.len || e.len > have)
            Corrupt();
        have -= e.len;
        emit(e.symbol);
    }
    if (produced != n_out)
        Corrupt();
}

// inverse of the two-point transforms: (average, difference) -> (a, b)
inline void Inverse14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b)
{
    const int hi = static_cast<int16_t>(h);
    const int ai = static_cast<int16_t>(l) + (hi & 1) + (hi >> 1);
    a = static_cast<uint16_t>(static_cast<int16_t>(ai));
    b = static_cast<uint16_t>(static_cast<int16_t>(ai - hi));
}
inline void Inverse16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b)
{
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xFFFF;
    const int aa = (d + bb - 0x8000) & 0xFFFF;
    b = static_cast<uint16_t>(bb);
    a = static_cast<uint16_t>(aa);
}

// in-place inverse 2-D wavelet over an nx x ny plane with element strides ox, oy
void InverseWavelet(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t max_value)
{
    const bool narrow = max_value < (1 << 14);
    auto inv = [narrow](uint16_t l, uint16_t h, uint16_t &a, uint16_t &b)
    { narrow ? Inverse14(l, h, a, b) : Inverse16(l, h, a, b); };
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n)
        p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    while (p >= 1)
    {
        uint16_t *py = in;
        uint16_t *const ey = in + static_cast<ptrdiff_t>(oy) * (ny - p2);
        const ptrdiff_t oy1 = static_cast<ptrdiff_t>(oy) * p, oy2 = static_cast<ptrdiff_t>(oy) * p2;
        const ptrdiff_t ox1 = static_cast<ptrdiff_t>(ox) * p, ox2 = static_cast<ptrdiff_t>(ox) * p2;
        for (; py <= ey; py += oy2)
        {
            uint16_t *px = py;
            uint16_t *const ex = py + static_cast<ptrdiff_t>(ox) * (nx - p2);
            for (; px <= ex; px += ox2)
            {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                uint16_t i00, i01, i10, i11;
                inv(*px, *p10, i00, i10);
                inv(*p01, *p11, i01, i11);
                inv(i00, i01, *px, *p01);
                inv(i10, i11, *p10, *p11);
            }
            if (nx & p)
            {
                uint16_t *p10 = px + oy1;
                uint16_t i00;
                inv(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p)
        {
            uint16_t *px = py;
            uint16_t *const ex = py + static_cast<ptrdiff_t>(ox) * (nx - p2);
            for (; px <= ex; px += ox2)
            {
                uint16_t *p01 = px + ox1;
                uint16_t i00;
                inv(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

} // namespace

void DecodePizBlock(const uint8_t *src, size_t n_src, const std::vector<int> &words_per_sample, int width, int lines,
                    uint8_t *dst)
{
    size_t total = 0;
    for (int w : words_per_sample)
        total += static_cast<size_t>(w) * width * lines;
    if (n_src < 4)
        Corrupt();
    uint16_t lo, hi;
    std::memcpy(&lo, src, 2);
    std::memcpy(&hi, src + 2, 2);
    size_t at = 4;
    constexpr int kBitmapBytes = 8192;
    std::vector<uint8_t> bitmap(kBitmapBytes, 0);
    if (hi >= kBitmapBytes)
        Corrupt();
    if (lo <= hi)
    {
        if (at + (hi - lo + 1) > n_src)
            Corrupt();
        std::memcpy(&bitmap[lo], src + at, hi - lo + 1);
        at += hi - lo + 1;
    }
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7))))
            lut[k++] = static_cast<uint16_t>(i);
    const uint16_t max_value = static_cast<uint16_t>(k - 1);
    if (at + 4 > n_src)
        Corrupt();
    int32_t length;
    std::memcpy(&length, src + at, 4);
    at += 4;
    if (length < 0 || at + static_cast<size_t>(length) > n_src)
        Corrupt();
    std::vector<uint16_t> plane(total);
    HuffmanDecode(src + at, static_cast<size_t>(length), plane.data(), total);
    // planes are stored channel after channel; multi-word samples interleave their words
    size_t start = 0;
    std::vector<size_t> starts;
    for (int w : words_per_sample)
    {
        starts.push_back(start);
        for (int j = 0; j < w; ++j)
            InverseWavelet(plane.data() + start + j, width, w, lines, width * w, max_value);
        start += static_cast<size_t>(w) * width * lines;
    }
    for (uint16_t &v : plane)
        v = lut[v];
    // scanline order: per line, per channel
    for (int y = 0; y < lines; ++y)
        for (size_t c = 0; c < words_per_sample.size(); ++c)
        {
            const size_t n = static_cast<size_t>(words_per_sample[c]) * width;
            std::memcpy(dst, plane.data() + starts[c] + n * y, n * 2);
            dst += n * 2;
        }
}

} // namespace mcpt
