// Float elementary functions that return, bit for bit, what GNU libc 2.35 (x86-64, the
// image's /lib/x86_64-linux-gnu/libm.so.6, FMA-capable host) returns.
//
// Why this exists.  The reference CPU integrator calls sinf / cosf / acosf / atan2f / atanf /
// tanf from the host's libm (reference src/utils/math.cpp:24-38,102-128,
// src/renderer/bsdfs/microfacet.cpp:21-38, src/renderer/emitters/spot_light.cpp,
// src/rtcore/primitives/{sphere,disk,cylinder}.cpp).  None of these is correctly rounded in
// glibc 2.35, so "a good sinf" on the device differs from the host's in the last bit on a few
// percent of the arguments, and one such bit at a bounce flips a later path decision: the GPU
// frame then matches the CPU frame only statistically.  With the host library's own algorithms
// restated here, device == host on every argument and the GPU frame is the CPU frame.
//
// Third-party dependency restated: GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11), not part of
// /root/reference.  Algorithms (published sources, restated, not copied):
//   sinf, cosf   sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c (the
//                "optimized routines" design: argument and polynomial in double, one multiply-
//                subtract range reduction below 120, a 192-bit 4/pi table above), in the
//                multiarch variant the dynamic loader selects on FMA hosts (products feeding a
//                sum are fused: sysdeps/x86_64/fpu/multiarch/s_sinf.c built with -mfma -mavx2);
//   acosf        sysdeps/ieee754/flt-32/e_acosf.c      (fdlibm, float arithmetic)
//   atanf        sysdeps/ieee754/flt-32/s_atanf.c      (fdlibm, float arithmetic)
//   atan2f       sysdeps/ieee754/flt-32/e_atan2f.c     (fdlibm, float arithmetic)
//   tanf         sysdeps/ieee754/flt-32/s_tanf.c (reduction shared with sinf, unfused), k_tanf.c (fdlibm)
// Pinned by tests/test_glibc_libm.py: every one of the 2^32 float arguments of the unary
// functions (2^31 + edge sweeps for atan2f) against the host's libm.
//
// errno, exception flags and signalling NaNs are not reproduced (the renderer never looks).
// Build with -ffp-contract=off: every operation below is the one written.
#ifndef MCPT_GLIBC_LIBM_H
#define MCPT_GLIBC_LIBM_H

#include <math.h>
#include <stdint.h>

#include "glibc_libm_tables.inc"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCPT_GL_HD __host__ __device__ __forceinline__
// (paths no render ever takes — |x| >= 120 in sinf / cosf — stay out of line: one copy per code object instead of ~250
//  instructions at each of the ~25 call sites of a full-feature kernel)
#define MCPT_GL_COLD __host__ __device__ inline __attribute__((noinline))
#else
#define MCPT_GL_HD inline
#define MCPT_GL_COLD inline
#endif

namespace mcpt
{
namespace gl
{

MCPT_GL_HD uint32_t bits(float x)
{
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    return u;
}
MCPT_GL_HD float from_bits(uint32_t u)
{
    float x;
    __builtin_memcpy(&x, &u, 4);
    return x;
}
MCPT_GL_HD double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- sinf / cosf ----------------------------------------------------------------------------
// Polynomials on [-pi/4, pi/4] in double: cos r = c0 + c1 r^2 + ... + c4 r^8,
// sin r = r + s1 r^3 + s2 r^5 + s3 r^7.
constexpr double kC0 = 0x1p0, kC1 = -0x1.ffffffd0c621cp-2, kC2 = 0x1.55553e1068f19p-5, kC3 = -0x1.6c087e89a359dp-10,
                 kC4 = 0x1.99343027bf8c3p-16;
constexpr double kS1 = -0x1.555545995a603p-3, kS2 = 0x1.1107605230bc4p-7, kS3 = -0x1.994eb3774cf24p-13;
constexpr double kHalfPiInvScaled = 0x1.45F306DC9C883p+23; // 2/pi * 2^24
constexpr double kHalfPi = 0x1.921FB54442D18p0;
constexpr double kPi63 = 0x1.921FB54442D18p-62; // 2 pi * 2^-64

MCPT_GL_HD uint32_t abstop12(float x) { return (bits(x) >> 20) & 0x7ffu; }

// `negate_cos`: quadrants 2 and 3 use the negated cosine polynomial.
MCPT_GL_HD float sincos_poly(double x, double x2, bool negate_cos, int n)
{
    if ((n & 1) == 0)
    {
        const double x3 = x * x2;
        const double s1 = fmad(x2, kS3, kS2);
        const double x7 = x3 * x2;
        const double s = fmad(x3, kS1, x);
        return static_cast<float>(fmad(x7, s1, s));
    }
    const double sg = negate_cos ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2 = fmad(x2, sg * kC4, sg * kC3);
    const double c1 = fmad(x2, sg * kC1, sg * kC0);
    const double x6 = x4 * x2;
    const double c = fmad(x4, sg * kC2, c1);
    return static_cast<float>(fmad(x6, c2, c));
}

// |x| < 120: one multiply-subtract.  The quadrant lands in bits 24..31 of the scaled product.
MCPT_GL_HD double reduce_fast(double x, int &n)
{
    const double r = x * kHalfPiInvScaled;
    n = (static_cast<int32_t>(r) + 0x800000) >> 24;
    return fmad(-static_cast<double>(n), kHalfPi, x);
}

// 4/pi, 192 bits, as overlapping 32-bit windows 8 bits apart: window k = bits [8k-24, 8k+8) of
// 0xa2f9836e4e441529fc2757d1f534ddc0db6295993c439041.
MCPT_GL_HD uint32_t inv_pio4_window(uint32_t k)
{
    const uint64_t w0 = 0xa2f9836e4e441529ull, w1 = 0xfc2757d1f534ddc0ull, w2 = 0xdb6295993c439041ull;
    // bit string B = w0 w1 w2 (192 bits, big endian); window k holds bits [8k - 24, 8k + 8) counted from
    // the top, with zeros above the string
    const int first = 8 * static_cast<int>(k) - 24; // index of the window's top bit
    uint64_t hi, lo;                                // the 128 bits starting at 64-bit word `word`
    const int word = first < 0 ? -1 : first / 64;
    if (word < 0)
        hi = 0, lo = w0;
    else if (word == 0)
        hi = w0, lo = w1;
    else if (word == 1)
        hi = w1, lo = w2;
    else
        hi = w2, lo = 0;
    const int off = first - 64 * word; // 0..63 within hi
    const uint64_t v = off == 0 ? hi : ((hi << off) | (lo >> (64 - off)));
    return static_cast<uint32_t>(v >> 32);
}

// |x| >= 120: exact fixed-point product with 4/pi.
MCPT_GL_HD double reduce_large(uint32_t xi, int &np)
{
    const uint32_t k = (xi >> 26) & 15u;
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = static_cast<uint32_t>(xi * inv_pio4_window(k));
    const uint64_t res1 = static_cast<uint64_t>(xi) * inv_pio4_window(k + 4);
    const uint64_t res2 = static_cast<uint64_t>(xi) * inv_pio4_window(k + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    const double x = static_cast<double>(static_cast<int64_t>(res0));
    np = static_cast<int>(n);
    return x * kPi63;
}

// |y| >= 120 (finite): the exact reduction.  Out of line, see MCPT_GL_COLD.
template <bool kCosine>
MCPT_GL_COLD float sin_or_cos_large(float y)
{
    const int flip = kCosine ? 1 : 0;
    const uint32_t xi = bits(y);
    const int sign = static_cast<int>(xi >> 31);
    int n;
    const double x = reduce_large(xi, n);
    const int q = n + sign;
    const double s = ((q + 1) & 2) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, (q & 2) != 0, n ^ flip);
}

// The statement of glibc's sinf / cosf has two ranges below 120: |y| < pi/4 (0.75, by the granularity of its comparison) without a
// reduction, and the rest with reduce_fast.  Here both run the second one's code: below 0.75 the scaled product is below 2^23 in
// magnitude, so the quadrant n is 0, reduce_fast returns fma(-0.0, pi/2, y) = y, the sign factor is 1 and the polynomial's
// arguments are the first range's, bit for bit (the sweep over all 2^32 arguments in the tests says so too) — lanes of one
// wavefront on either side of 0.75 then share one polynomial evaluation instead of executing two.
template <bool kCosine>
MCPT_GL_HD float sin_or_cos(float y)
{
    const int flip = kCosine ? 1 : 0;
    if (abstop12(y) < abstop12(120.0f))
    {
        int n;
        const double x = reduce_fast(static_cast<double>(y), n);
        const double s = ((n + 1) & 2) ? -1.0 : 1.0; // {1, -1, -1, 1}[n & 3]
        const float r = sincos_poly(x * s, x * x, (n & 2) != 0, n ^ flip);
        return abstop12(y) < abstop12(0x1p-12f) ? (kCosine ? 1.0f : y) : r;
    }
    if (abstop12(y) < abstop12(__builtin_inff()))
        return sin_or_cos_large<kCosine>(y);
    return __builtin_nanf("");
}

MCPT_GL_HD float sinf(float y) { return sin_or_cos<false>(y); }
MCPT_GL_HD float cosf(float y) { return sin_or_cos<true>(y); }

// ---- acosf ----------------------------------------------------------------------------------
MCPT_GL_HD float acos_ratio(float z)
{
    constexpr float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                    pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
                    qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}

MCPT_GL_HD float acosf(float x)
{
    constexpr float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const int32_t hx = static_cast<int32_t>(bits(x)), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000)
        return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000)
        return __builtin_nanf("");
    // e_acosf.c has three ranges, each with its own rational p(z) / q(z) of a range-specific z: |x| < 0.5 (z = x^2), x < -0.5
    // (z = (1 + x) / 2) and x > 0.5 (z = (1 - x) / 2, with sqrt z split into head and tail).  The ratio and the square root are
    // evaluated ONCE here, on the z of the lane's range: the same operations on the same operands, once per wavefront instead of
    // once per range present in it.
    const bool small = ix < 0x3f000000, negative = hx < 0;
    if (small && ix <= 0x32800000)
        return pio2_hi + pio2_lo;
    const float z = small ? x * x : (negative ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f);
    const float r = acos_ratio(z);
    if (small)
        return pio2_hi - (x - (pio2_lo - x * r));
    const float s = ::sqrtf(z);
    if (negative) // x < -0.5
    {
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float df = from_bits(bits(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// ---- atanf ----------------------------------------------------------------------------------
MCPT_GL_HD float atanf(float x)
{
    constexpr float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    constexpr float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    constexpr float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f, a3 = -1.1111110449e-01f,
                    a4 = 9.0908870101e-02f, a5 = -7.6918758452e-02f, a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f,
                    a8 = 4.9768779427e-02f, a9 = -3.6531571299e-02f, a10 = 1.6285819933e-02f;
    const int32_t hx = static_cast<int32_t>(bits(x)), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) // |x| >= 2^25
    {
        if (ix > 0x7f800000)
            return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x3ee00000) // |x| < 0.4375
    {
        if (ix < 0x31000000)
            return x;
        id = -1;
    }
    else
    {
        // (s_atanf.c divides in each of its four ranges; numerator and denominator are picked per range here and divided once)
        x = ::fabsf(x);
        float num, den;
        if (ix < 0x3f980000)
        {
            if (ix < 0x3f300000)
                id = 0, num = 2.0f * x - 1.0f, den = 2.0f + x;
            else
                id = 1, num = x - 1.0f, den = x + 1.0f;
        }
        else
        {
            if (ix < 0x401c0000)
                id = 2, num = x - 1.5f, den = 1.0f + 1.5f * x;
            else
                id = 3, num = -1.0f, den = x;
        }
        x = num / den;
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0)
        return x - x * (s1 + s2);
    const float hi = id == 0 ? hi0 : (id == 1 ? hi1 : (id == 2 ? hi2 : hi3));
    const float lo = id == 0 ? lo0 : (id == 1 ? lo1 : (id == 2 ? lo2 : lo3));
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

// ---- atan2f ---------------------------------------------------------------------------------
MCPT_GL_HD float atan2f(float y, float x)
{
    constexpr float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                    pi_lo = -8.7422776573e-08f;
    const int32_t hx = static_cast<int32_t>(bits(x)), ix = hx & 0x7fffffff;
    const int32_t hy = static_cast<int32_t>(bits(y)), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000)
        return x + y;
    if (hx == 0x3f800000)
        return gl::atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0)
    {
        if (m < 2)
            return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000)
    {
        if (iy == 0x7f800000)
        {
            switch (m)
            {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m)
        {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = gl::atanf(::fabsf(y / x));
    switch (m)
    {
    case 0: return z;
    case 1: return from_bits(bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ---- tanf -------------------------------------------------------------------------------------
MCPT_GL_HD float kernel_tanf(float x, float y, int iy)
{
    constexpr float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    constexpr float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f,
                    T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f,
                    T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f,
                    T12 = 2.5907305826e-05f;
    const int32_t hx = static_cast<int32_t>(bits(x)), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) // |x| < 2^-13
    {
        if (static_cast<int>(x) == 0)
        {
            if ((ix | (iy + 1)) == 0)
                return 1.0f / ::fabsf(x);
            if (iy == 1)
                return x;
            return -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) // |x| >= 0.6744
    {
        if (hx < 0)
            x = -x, y = -y;
        const float z = pio4 - x, w = pio4lo - y;
        x = z + w, y = 0.0f;
        if (::fabsf(x) < 0x1p-13f)
            return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
    }
    float z = x * x, w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140)
    {
        v = static_cast<float>(iy);
        return static_cast<float>(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1)
        return w;
    // -1 / (x + r), accurately
    z = from_bits(bits(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = from_bits(bits(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}

// s_tanf.c: the argument reduction of sinf / cosf (in double; this translation unit of the library is
// not built with FMA, so the multiply-subtract is two operations), the remainder split into a float
// head and tail for the fdlibm kernel.
MCPT_GL_HD float tanf(float x)
{
    const int32_t ix = static_cast<int32_t>(bits(x)) & 0x7fffffff;
    if (ix <= 0x3f490fda)
        return kernel_tanf(x, 0.0f, 1);
    if (ix >= 0x7f800000)
        return __builtin_nanf("");
    double dx = x;
    int n;
    if (abstop12(x) < abstop12(120.0f))
    {
        const double r = dx * kHalfPiInvScaled;
        n = (static_cast<int32_t>(r) + 0x800000) >> 24;
        dx = dx - static_cast<double>(n) * kHalfPi;
    }
    else
    {
        const uint32_t xi = bits(x);
        dx = reduce_large(xi, n);
        dx = (xi >> 31) ? -dx : dx;
    }
    const float y0 = static_cast<float>(dx), y1 = static_cast<float>(dx - static_cast<double>(y0));
    return kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

// ---- exp, log in double (the medium code) ---------------------------------------------------------
// The reference's homogeneous medium calls the C library's DOUBLE exp / log on float arguments
// (src/renderer/medium/homogeneous.cpp:18-24,37,45,58: unqualified `exp(...)` / `log(...)` resolve to
// ::exp(double) / ::log(double)) and rounds the result to float.  glibc 2.35's exp / log are not
// correctly rounded (0.51 ulp), so a device library's exp / log can differ in the last bit of the double,
// which very rarely — about once in 2^29 calls — changes the float.  Restated: sysdeps/ieee754/dbl-64/e_exp.c,
// e_log.c (the "optimized routines" design: 128-entry tables, see glibc_libm_tables.inc), in the multiarch
// variants __exp_fma / __log_fma the loader selects on FMA hosts.  Which products are fused was read off
// that library's machine code (scratch disassembly of libm.so.6; noted per line below) — the C sources
// leave it to the compiler.  Pinned by tests/test_glibc_libm.py: all 2^32 float arguments (as doubles) and
// 2^28 random double bit patterns against the host, on the host build and on the device.
MCPT_GL_HD uint64_t bits64(double x)
{
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
MCPT_GL_HD double from_bits64(uint64_t u)
{
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ const uint64_t kExpTab[256] = {MCPT_GLIBC_EXP_TAB};
__device__ const double kLogTab[256] = {MCPT_GLIBC_LOG_TAB};
#else
static const uint64_t kExpTab[256] = {MCPT_GLIBC_EXP_TAB};
static const double kLogTab[256] = {MCPT_GLIBC_LOG_TAB};
#endif

// e_exp.c specialcase(): the scale factor would over- / underflow.
MCPT_GL_HD double exp_special(double tmp, uint64_t sbits, uint64_t ki)
{
    if ((ki & 0x80000000ull) == 0)
    {
        // k > 0: the exponent of scale might have overflowed by <= 460
        sbits -= 1009ull << 52;
        const double scale = from_bits64(sbits);
        return 0x1p1009 * fmad(scale, tmp, scale);
    }
    // k < 0: careful in the subnormal range
    sbits += 1022ull << 52;
    const double scale = from_bits64(sbits);
    const double st = scale * tmp; // (not fused in __exp_fma: the product is used twice)
    double y = scale + st;
    if (y < 1.0)
    {
        // round y to the right precision before scaling it into the subnormal range
        double lo = scale - y + st;
        const double hi = 1.0 + y;
        lo = 1.0 - hi + y + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0)
            y = 0.0; // no -0
    }
    return 0x1p-1022 * y;
}

MCPT_GL_HD double exp(double x)
{
    constexpr double kInvLn2N = 0x1.71547652b82fep0 * 128, kShift = 0x1.8p52, kNegLn2hiN = -0x1.62e42fefa0000p-8,
                     kNegLn2loN = -0x1.cf79abc9e3b3ap-47;
    constexpr double kC2 = 0x1.ffffffffffdbdp-2, kC3 = 0x1.555555555543cp-3, kC4 = 0x1.55555cf172b91p-5, kC5 = 0x1.1111167a4d017p-7;
    uint32_t abstop = static_cast<uint32_t>(bits64(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) // |x| < 2^-54 or |x| >= 512
    {
        if (abstop - 0x3c9u >= 0x80000000u)
            return 1.0 + x; // tiny
        if (abstop >= 0x409u) // |x| >= 1024, inf, nan
        {
            if (bits64(x) == bits64(-__builtin_inf()))
                return 0.0;
            if (abstop >= 0x7ffu)
                return 1.0 + x;
            return (bits64(x) >> 63) ? 0.0 : __builtin_inf(); // __math_uflow(0) / __math_oflow(0)
        }
        abstop = 0; // large x is special cased below
    }
    // exp(x) = 2^(k/N) * exp(r), r = x - k ln2/N in [-ln2/2N, ln2/2N]
    double kd = fmad(x, kInvLn2N, kShift); // fused in __exp_fma
    const uint64_t ki = bits64(kd);
    kd -= kShift;
    double r = fmad(kd, kNegLn2hiN, x);
    r = fmad(kd, kNegLn2loN, r);
    const uint64_t idx = 2 * (ki % 128);
    const uint64_t top = ki << 45;
    const double tail = from_bits64(kExpTab[idx]);
    const uint64_t sbits = kExpTab[idx + 1] + top;
    const double r2 = r * r;
    // tmp = tail + r + r2 (C2 + r C3) + r2 r2 (C4 + r C5), as __exp_fma evaluates it
    double tmp = fmad(fmad(r, kC3, kC2), r2, tail + r);
    tmp = fmad(r2 * r2, fmad(r, kC5, kC4), tmp);
    if (abstop == 0)
        return exp_special(tmp, sbits, ki);
    const double scale = from_bits64(sbits);
    return fmad(scale, tmp, scale);
}

MCPT_GL_HD double log(double x)
{
    constexpr double kLn2hi = 0x1.62e42fefa3800p-1, kLn2lo = 0x1.ef35793c76730p-45;
    constexpr double kA0 = -0x1.0000000000001p-1, kA1 = 0x1.555555551305bp-2, kA2 = -0x1.fffffffeb459p-3, kA3 = 0x1.999b324f10111p-3,
                     kA4 = -0x1.55575e506c89fp-3;
    constexpr double kB0 = -0x1p-1, kB1 = 0x1.5555555555577p-2, kB2 = -0x1.ffffffffffdcbp-3, kB3 = 0x1.999999995dd0cp-3,
                     kB4 = -0x1.55555556745a7p-3, kB5 = 0x1.24924a344de3p-3, kB6 = -0x1.fffffa4423d65p-4, kB7 = 0x1.c7184282ad6cap-4,
                     kB8 = -0x1.999eb43b068ffp-4, kB9 = 0x1.78182f7afd085p-4, kB10 = -0x1.5521375d145cdp-4;
    uint64_t ix = bits64(x);
    const uint32_t top = static_cast<uint32_t>(ix >> 48);
    constexpr uint64_t kLo = 0x3fee000000000000ull, kHi = 0x3ff1090000000000ull; // 1 - 2^-4, 1 + 0x1.09p-4
    if (ix - kLo < kHi - kLo)
    {
        // close to 1: a polynomial in r = x - 1 with the leading terms in double-double
        if (ix == 0x3ff0000000000000ull)
            return 0.0;
        const double r = x - 1.0, r2 = r * r, r3 = r * r2;
        const double q0 = fmad(r2, kB3, fmad(r, kB2, kB1));
        const double q1 = fmad(r2, kB6, fmad(r, kB5, kB4));
        double q2 = fmad(r2, kB9, fmad(r, kB8, kB7));
        q2 = fmad(r3, kB10, q2);
        const double p = fmad(fmad(q2, r3, q1), r3, q0); // y = r3 * p, folded into the last sum below
        const double w = r * 0x1p27;
        const double rhi = r + w - w, rlo = r - rhi;
        const double ww = rhi * rhi * kB0;
        const double hi = r + ww;
        double lo = r - hi + ww;
        lo = fmad(kB0 * rlo, rhi + r, lo); // fused in __log_fma
        return hi + fmad(p, r3, lo);       // y = r3 p + lo (fused), + hi
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u)
    {
        // x < 2^-1022, infinite or NaN
        if (ix * 2 == 0)
            return -__builtin_inf();
        if (ix == bits64(__builtin_inf()))
            return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u)
            return __builtin_nan("");
        ix = bits64(x * 0x1p52); // subnormal: normalise
        ix -= 52ull << 52;
    }
    // x = 2^k z, z in [0.6875, 1.375) cut into 128 subintervals; log x = log1p(z / c - 1) + log c + k ln 2
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const uint32_t i = static_cast<uint32_t>(tmp >> 45) % 128;
    const int64_t k = static_cast<int64_t>(tmp) >> 52;
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = kLogTab[2 * i], logc = kLogTab[2 * i + 1];
    const double z = from_bits64(iz);
    const double r = fmad(z, invc, -1.0); // __FP_FAST_FMA form
    const double kd = static_cast<double>(static_cast<int32_t>(k));
    const double w = fmad(kd, kLn2hi, logc);
    const double hi = w + r;
    const double lo = fmad(kd, kLn2lo, w - hi + r);
    const double r2 = r * r;
    // y = lo + r2 A0 + r r2 (A1 + r A2 + r2 (A3 + r A4)) + hi
    const double q = fmad(fmad(r, kA4, kA3), r2, fmad(r, kA2, kA1));
    return fmad(r * r2, q, fmad(r2, kA0, lo)) + hi;
}

} // namespace gl
} // namespace mcpt

#endif // MCPT_GLIBC_LIBM_H
