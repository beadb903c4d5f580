// Small vector / matrix / sampling vocabulary of the renderer, usable from
// host code (scene commit) and from HIP device code (kernels).
//
// Numerical contract (it decides parity with the reference CPU integrator):
//   * every operation is a single IEEE float32 operation in the order written;
//     build with -ffp-contract=off, never with fast-math;
//   * `v / s` and `v / w` multiply by reciprocals, normalisation is
//     v * (1 / |v|), direction transforms normalise their result — as the
//     reference's tensor library does (src/tensor/vec3.cpp:107-185,
//     mat4.cpp:264-273, vec4.hpp:55);
//   * where the reference evaluates a sub-expression in double because only
//     the double libm overload is visible, the double is
//     written out here too (see D(), the Pow3d/Pow5d helpers and callers).
#ifndef MCPT_VECMATH_H
#define MCPT_VECMATH_H

#include <math.h>
#include <stdint.h>

#include "device_scene.h"
#include "glibc_libm.h"

#if defined(__HIPCC__)
#define MCPT_HD __host__ __device__ __forceinline__
#define MCPT_HD_ATTR __host__ __device__
#else
#define MCPT_HD inline
#define MCPT_HD_ATTR
#endif
// Large bodies with many call sites (direction -> spherical angles, varying textures): ONE copy per code object instead of one per
// call site where MCPT_OUTLINE is set — the instantiations with every BSDF model and an environment map are 150-220 KB of code
// against a 64 KB instruction cache, and 0.9 % of their instruction fetches miss (EXPERIMENTS R5-9).
#ifndef MCPT_OUTLINE
#define MCPT_OUTLINE 0
#endif
#if MCPT_OUTLINE
#define MCPT_COLD MCPT_HD_ATTR inline __attribute__((noinline))
#else
#define MCPT_COLD MCPT_HD
#endif

namespace mcpt
{

constexpr float kEpsFloat = 1.1920928955078125e-07f; // FLT_EPSILON, defs.hpp:24
constexpr float kEpsDistance = 1e-4f;                // defs.hpp:25
constexpr float kEps = 0.01f;                        // defs.hpp:26
constexpr float kMaxFloat = 3.402823466e+38f;
constexpr float kLowestFloat = -3.402823466e+38f;

constexpr float kPi = 3.141592653589793f; // math.hpp:17-23
constexpr float k2Pi = 3.141592653589793f * 2.0f;
constexpr float kPiDiv2 = 3.141592653589793f * 0.5f;
constexpr float kPiDiv4 = 3.141592653589793f * 0.25f;
constexpr float k1DivPi = 1.0f / kPi;
constexpr float k1Div2Pi = 1.0f / k2Pi;
constexpr float k1Div4Pi = 1.0f / (4.0f * kPi);

MCPT_HD double D(float x) { return static_cast<double>(x); }

struct V2
{
    float u, v;
};
MCPT_HD V2 mk2(float u, float v) { return V2{u, v}; }
MCPT_HD V2 operator+(V2 a, V2 b) { return V2{a.u + b.u, a.v + b.v}; }
MCPT_HD V2 operator-(V2 a, V2 b) { return V2{a.u - b.u, a.v - b.v}; }
MCPT_HD V2 operator*(float t, V2 a) { return V2{t * a.u, t * a.v}; }

struct V3
{
    float x, y, z;
};
MCPT_HD V3 mk3(float x, float y, float z) { return V3{x, y, z}; }
MCPT_HD V3 splat(float s) { return V3{s, s, s}; }
MCPT_HD V3 from(const Vec3f &v) { return V3{v.x, v.y, v.z}; }
MCPT_HD V3 xyz(const float4 &v) { return V3{v.x, v.y, v.z}; }
MCPT_HD float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

MCPT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
MCPT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MCPT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MCPT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
MCPT_HD V3 operator/(V3 a, V3 b)
{
    const float k0 = 1.0f / b.x, k1 = 1.0f / b.y, k2 = 1.0f / b.z;
    return V3{a.x * k0, a.y * k1, a.z * k2};
}
MCPT_HD V3 operator+(V3 a, float t) { return V3{a.x + t, a.y + t, a.z + t}; }
MCPT_HD V3 operator-(V3 a, float t) { return V3{a.x - t, a.y - t, a.z - t}; }
MCPT_HD V3 operator*(V3 a, float t) { return V3{a.x * t, a.y * t, a.z * t}; }
MCPT_HD V3 operator/(V3 a, float t)
{
    const float k = 1.0f / t;
    return V3{a.x * k, a.y * k, a.z * k};
}
MCPT_HD V3 operator+(float t, V3 a) { return V3{t + a.x, t + a.y, t + a.z}; }
MCPT_HD V3 operator-(float t, V3 a) { return V3{t - a.x, t - a.y, t - a.z}; }
MCPT_HD V3 operator*(float t, V3 a) { return V3{t * a.x, t * a.y, t * a.z}; }
MCPT_HD void operator+=(V3 &a, V3 b) { a = a + b; }
MCPT_HD void operator*=(V3 &a, V3 b) { a = a * b; }
MCPT_HD void operator*=(V3 &a, float t) { a = a * t; }

MCPT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MCPT_HD float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
MCPT_HD V3 normalize(V3 a)
{
    const float k = 1.0f / length(a);
    return a * k;
}
MCPT_HD V3 cross(V3 a, V3 b) // vec3.cpp:192-196, including its -x*z + z*x form
{
    return V3{a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x};
}
MCPT_HD V3 vmin(V3 a, V3 b) { return V3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
MCPT_HD V3 vmax(V3 a, V3 b) { return V3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
MCPT_HD V3 vsqrt(V3 a) { return V3{sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
MCPT_HD float sqr(float t) { return t * t; }
MCPT_HD V3 sqr(V3 t) { return t * t; }
MCPT_HD float max_component(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

MCPT_HD float lerp(float a, float b, float t) { return (1.0f - t) * a + t * b; } // math.hpp:74-77
MCPT_HD V3 lerp(V3 a, V3 b, float t) { return (1.0f - t) * a + t * b; }

// ---- matrices -------------------------------------------------------------
MCPT_HD float dot4(const float *a, float bx, float by, float bz, float bw) // vec4.cpp:164-167
{
    return a[0] * bx + a[1] * by + a[2] * bz + a[3] * bw;
}

MCPT_HD V3 transform_point(const Mat4f &a, V3 p) // mat4.cpp:264-267, vec4.cpp:93-97
{
    const float x = dot4(a.m + 0, p.x, p.y, p.z, 1.0f), y = dot4(a.m + 4, p.x, p.y, p.z, 1.0f),
                z = dot4(a.m + 8, p.x, p.y, p.z, 1.0f), w = dot4(a.m + 12, p.x, p.y, p.z, 1.0f);
    const float k = 1.0f / w;
    return V3{x * k, y * k, z * k};
}

MCPT_HD V3 transform_dir(const Mat4f &a, V3 d) // mat4.cpp:270-273: normalised
{
    return normalize(V3{dot4(a.m + 0, d.x, d.y, d.z, 0.0f), dot4(a.m + 4, d.x, d.y, d.z, 0.0f),
                        dot4(a.m + 8, d.x, d.y, d.z, 0.0f)});
}

// ---- RNG and low-discrepancy points (math.hpp:29-63) -----------------------
MCPT_HD uint32_t tea4(uint32_t v0, uint32_t v1)
{
    uint32_t sum = 0;
#pragma unroll
    for (int round = 0; round < 4; ++round)
    {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

// ---- low-discrepancy build (throughput mode 2 of mcpt_renderer_set_rng; no reference counterpart) ----------------------
// The random stream is ONE 32-bit word handed by reference to every sampling function (the reference's shape,
// math.hpp:43-63).  A translation unit compiled with MCPT_LOW_DISCREPANCY (hip/render_variants_lowdisc.hip; the host twin
// tests/emu/libmcpt_emu_ld.so also defines MCPT_LOW_DISCREPANCY_HOST) gives that word another meaning — no call site changes:
//     bits 31..19  s = index of the sample within its pixel (at most 8192 spp)
//     bits 18..7   p = 12 bits of a PCG hash of (seed, pixel): which of 4096 scrambles the pixel uses
//     bits  6..0   d = draws made so far in this sample (mod 128) = the dimension; draw 128 carries into p: the sample goes
//                      on with the next scramble (fresh pair seeds), 2^19 draws before anything repeats
// and a draw returns an Owen-scrambled Sobol point: dimensions come in pairs (d >> 1), every pair is the first two Sobol
// dimensions — a (0,2)-sequence in base 2 — over an Owen-shuffled sample index, each dimension Owen-scrambled with its own
// seed ("padded" (0,2)-sequences: Kollig & Keller 2002; the hash-based nested uniform scramble is Burley 2020 / Laine &
// Karras 2011).  Every pair of draws of a pixel is therefore stratified over ANY power-of-two run of its samples, and pairs
// are independent of each other and (up to the 4096 scrambles) across pixels.
#if defined(MCPT_LOW_DISCREPANCY) && (defined(__HIP_DEVICE_COMPILE__) || defined(MCPT_LOW_DISCREPANCY_HOST))
#define MCPT_LOW_DISCREPANCY_ACTIVE 1
#else
#define MCPT_LOW_DISCREPANCY_ACTIVE 0
#endif

MCPT_HD uint32_t ld_reverse_bits(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    return ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
#endif
}

// Laine-Karras permutation: a bijection whose output bit k depends on input bits <= k only; between two bit reversals it is a
// nested uniform (Owen) scramble in base 2.
MCPT_HD uint32_t ld_laine_karras(uint32_t x, uint32_t seed)
{
    x += seed;
    x ^= x * 0x6c50b47cu;
    x ^= x * 0xb82f1e52u;
    x ^= x * 0xc7afe638u;
    x ^= x * 0x8d22f6e6u;
    return x;
}

// Second Sobol dimension, bit-reversed: its generator matrix is Pascal's triangle mod 2 (direction numbers v_0 = 2^31,
// v_k = v_(k-1) ^ (v_(k-1) >> 1)), i.e. bit j of the reversed point = XOR of the index bits k that contain j as a bit
// subset — a superset-sum butterfly over GF(2).
MCPT_HD uint32_t ld_sobol1_reversed(uint32_t index)
{
    index ^= (index >> 1) & 0x55555555u;
    index ^= (index >> 2) & 0x33333333u;
    index ^= (index >> 4) & 0x0f0f0f0fu;
    index ^= (index >> 8) & 0x00ff00ffu;
    index ^= (index >> 16);
    return index;
}

MCPT_HD uint32_t ld_pack(uint32_t sample, uint32_t pixel_hash) { return (sample << 19) | ((pixel_hash >> 20) << 7); }

MCPT_HD float ld_next(uint32_t &word)
{
    const uint32_t d = word & 0x7fu, s = word >> 19, p = (word >> 7) & 0xfffu;
    // (the increment carries from the 7 dimension bits into the 12 scramble bits — never into the sample's: after 128 draws a
    //  sample goes on with the NEXT scramble's pair seeds instead of repeating its own first 128 numbers, which biased paths
    //  longer than about 12 vertices: volumetric-caustic, chains through glass — round 4's advisor)
    word = (word & 0xfff80000u) | ((word + 1u) & 0x0007ffffu);
    // (pcg_hash below; written out here because it is declared after lcg_next)
    const uint32_t state = ((d >> 1) * 4096u + p) * 747796405u + 2891336453u;
    const uint32_t mixed = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    const uint32_t pair_seed = (mixed >> 22u) ^ mixed;
    // Owen-shuffled sample index: scrambling the index in base 2 permutes every aligned power-of-two block within itself
    const uint32_t index = ld_reverse_bits(ld_laine_karras(ld_reverse_bits(s), pair_seed));
    // the point's bits, reversed (dimension 0 = van der Corput: the reversed point IS the index), scrambled, reversed back
    const uint32_t reversed = (d & 1u) ? ld_sobol1_reversed(index) : index;
    const uint32_t x = ld_reverse_bits(ld_laine_karras(reversed, pair_seed * 0x9e3779b9u + 0x7f4a7c15u + (d & 1u) * 0x632be5abu));
    return static_cast<float>(x >> 8) / static_cast<float>(0x01000000u);
}

MCPT_HD float lcg_next(uint32_t &state)
{
#if MCPT_LOW_DISCREPANCY_ACTIVE
    return ld_next(state);
#else
    state = state * 1664525u + 1013904223u;
    return static_cast<float>(state & 0x00ffffffu) / static_cast<float>(0x01000000u);
#endif
}

// PCG output hash (O'Neill's RXS-M-XS 32-bit permutation over one LCG step): seeds of the independent-sample
// RNG mode (no reference counterpart; the reference threads one LCG through all samples of a pixel).
MCPT_HD uint32_t pcg_hash(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// Base-2 radical inverse with the reference's float index update
// (`index *= base_inv`, math.hpp:36-38).
MCPT_HD float radical_inverse2(uint32_t index)
{
    float result = 0.0f, frac = 0.5f;
    while (index > 0)
    {
        result += frac * static_cast<float>(index % 2u);
        index = static_cast<uint32_t>(static_cast<float>(index) * 0.5f);
        frac *= 0.5f;
    }
    return result;
}

MCPT_HD float power_heuristic(float a, float b) // math.cpp:8-13
{
    a *= a;
    b *= b;
    return a / (a + b);
}

// Exact integer powers in double standing in for the reference's
// pow(float, int) calls (double libm pow).  x*x*x in double differs from a
// correctly rounded pow by < 1 ulp(double); after the cast to float the two
// agree except on ~1e-9 of inputs.
MCPT_HD double pow2d(float x) { return D(x) * D(x); }
MCPT_HD double pow3d(float x) { return D(x) * D(x) * D(x); }
MCPT_HD double pow5d(float x)
{
    const double x2 = D(x) * D(x);
    return x2 * x2 * D(x);
}

MCPT_HD void sample_hemisphere_cosine(float xi0, float xi1, V3 &dir, float &pdf) // math.cpp:31-38
{
    const float cos_t = sqrtf(xi0), phi = k2Pi * xi1;
    const float sin_t = sqrtf(1.0f - sqr(cos_t)); // double sqrt -> float == sqrtf
    dir = V3{sin_t * gl::cosf(phi), sin_t * gl::sinf(phi), cos_t};
    pdf = k1DivPi * cos_t;
}

MCPT_HD V3 sample_sphere_uniform(float xi0, float xi1) // math.cpp:24-29
{
    const float cos_t = 1.0f - 2.0f * xi0, phi = k2Pi * xi1;
    const float sin_t = sqrtf(1.0f - sqr(cos_t));
    return V3{sin_t * gl::cosf(phi), sin_t * gl::sinf(phi), cos_t};
}

MCPT_HD V3 sample_cone_uniform(float cos_cutoff, float xi0, float xi1) // math.cpp:15-22
{
    const float cos_t = 1.0f - (1.0f - cos_cutoff) * xi0, phi = 2.0f * kPi * xi1;
    const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
    return V3{sin_t * gl::cosf(phi), sin_t * gl::sinf(phi), cos_t};
}

MCPT_HD uint32_t cdf_search(uint32_t num, const float *cdf, float target) // math.cpp:40-55
{
    uint32_t lo = 0, hi = num;
    while (lo + 1 != hi)
    {
        const uint32_t mid = (lo + hi) >> 1;
        const float c = cdf[mid];
        if (c < target)
            lo = mid;
        else if (c > target)
            hi = mid;
        else
            return mid;
    }
    return hi;
}

// cdf_search for LONG tables in memory (an environment map's row and column tables: 10 + 11 levels, each a load that waits for the
// one before).  The same bisection, the same comparisons on the same entries in the same order, hence the same index for ANY table
// content (quirk Q7 makes the reference search tables that are not even monotone) — but several levels per round trip: the entries at
// the midpoint and at the possible next midpoints are loaded together (2^levels - 1 independent loads, all inside [lo, hi)), then those
// levels are decided from registers.  Measured on matpreview (issue-bound kernels: waiting less buys little, selecting among more
// loaded entries costs instructions), rough dielectric / rough conductor, ms: 1 level per round 674.9 / 419.4, **2: 673.0 / 418.5**,
// 3: 679.5 / 423.6, 4: 703 / 433 (profiles/r05_experiments/envmap_bisection_levels_per_round_ab.json).
#ifndef MCPT_CDF_LEVELS
#define MCPT_CDF_LEVELS 2
#endif
MCPT_HD uint32_t cdf_search_rounds(uint32_t num, const float *cdf, float target)
{
    // kLevels levels per round trip: the bisection tree below (lo, hi) in heap order, node k's children 2k and 2k + 1
    constexpr uint32_t kLevels = MCPT_CDF_LEVELS, kNodes = 1u << kLevels;
    uint32_t lo = 0, hi = num;
    while (lo + 1 != hi)
    {
        uint32_t l[kNodes], h[kNodes], m[kNodes];
        float c[kNodes];
        l[1] = lo, h[1] = hi;
#pragma unroll
        for (uint32_t k = 1; k < kNodes; ++k)
        {
            m[k] = (l[k] + h[k]) >> 1;
            if (2 * k + 1 < kNodes)
                l[2 * k] = l[k], h[2 * k] = m[k], l[2 * k + 1] = m[k], h[2 * k + 1] = h[k];
        }
#pragma unroll
        for (uint32_t k = 1; k < kNodes; ++k)
            c[k] = cdf[m[k]]; // (independent loads, every index inside [lo, hi): a degenerate interval's midpoint is its lower end)
        uint32_t k = 1;
#pragma unroll
        for (uint32_t level = 0; level < kLevels; ++level)
        {
            if (level != 0 && lo + 1 == hi)
                return hi;
            // entry and midpoint of node k (k is one of 2^level values: selects, no indexed registers)
            float ck = c[1u << level];
            uint32_t mk = m[1u << level];
#pragma unroll
            for (uint32_t n = (1u << level) + 1; n < (2u << level); ++n)
                ck = k == n ? c[n] : ck, mk = k == n ? m[n] : mk;
            if (!(ck < target) && !(ck > target))
                return mk;
            const bool right = ck < target;
            lo = right ? mk : lo, hi = right ? hi : mk;
            k = 2 * k + (right ? 1u : 0u);
        }
    }
    return hi;
}
MCPT_HD uint32_t cdf_search_long(uint32_t num, const float *cdf, float target)
{
#if MCPT_WAVE_DEVICE
    return cdf_search_rounds(num, cdf, target);
#else
    return cdf_search(num, cdf, target); // (host threads: the cache holds the table's top levels, the plain loop reads less)
#endif
}

MCPT_HD bool solve_quadratic(float a, float b, float c, float &x0, float &x1) // math.cpp:57-98
{
    if (a == 0.0f)
    {
        if (b != 0.0f)
        {
            x0 = x1 = -c / b;
            return true;
        }
        return false;
    }
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f)
        return false;
    const float root = sqrtf(disc);
    const float q = (b < 0.0f) ? -0.5f * (b - root) : -0.5f * (b + root);
    x0 = q / a;
    x1 = c / q;
    if (x0 > x1)
    {
        const float t = x0;
        x0 = x1;
        x1 = t;
    }
    return true;
}

// y-up spherical coordinates (math.cpp:101-128)
struct Spherical
{
    float theta, phi;
};
MCPT_COLD Spherical spherical_of(V3 v)
{
    Spherical s;
    v = normalize(v);
    s.theta = gl::acosf(fminf(1.0f, fmaxf(-1.0f, v.y)));
    if (v.z == 0 && v.x == 0)
    {
        s.phi = 0;
    }
    else
    {
        s.phi = gl::atan2f(v.z, v.x);
        if (s.phi < 0.0f)
            s.phi += 2.0f * kPi;
    }
    return s;
}
MCPT_HD void to_spherical(V3 v, float &theta, float &phi)
{
    const Spherical s = spherical_of(v);
    theta = s.theta, phi = s.phi;
}

MCPT_HD V3 from_spherical(float theta, float phi, float r)
{
    const float sin_t = gl::sinf(theta);
    return V3{r * gl::sinf(phi) * sin_t, r * gl::cosf(theta), r * gl::cosf(phi) * sin_t};
}

// math.cpp:130-145; the reciprocal square roots are float / double-sqrt.
MCPT_HD V3 frame_to_world(V3 local, V3 up)
{
    V3 c;
    if (sqrt(D(sqr(up.x) + sqr(up.z))) > D(kEpsFloat))
    {
        const float k = static_cast<float>(1.0 / sqrt(D(sqr(up.x) + sqr(up.z))));
        c = V3{up.z * k, 0, -up.x * k};
    }
    else
    {
        const float k = static_cast<float>(1.0 / sqrt(D(sqr(up.y) + sqr(up.z))));
        c = V3{0, up.z * k, -up.y * k};
    }
    const V3 b = normalize(cross(c, up));
    return normalize(local.x * b + local.y * c + local.z * up);
}

MCPT_HD V3 reflect(V3 wi, V3 n) // ray.cpp:49-52
{
    return normalize(wi - 2.0f * dot(wi, n) * n);
}

MCPT_HD bool refract(V3 wi, V3 n, float eta_inv, V3 &wt) // ray.cpp:54-68
{
    const float cos_t = fabsf(dot(wi, n));
    const float k = 1.0f - sqr(eta_inv) * (1.0f - sqr(cos_t));
    if (k < 0)
        return false;
    wt = normalize(eta_inv * wi + (eta_inv * cos_t - sqrtf(k)) * n);
    return true;
}

// ---- GGX pieces shared by the LUT build (host) and the BSDFs (device) ------
MCPT_HD void ggx_sample_iso(float xi0, float xi1, float alpha, V3 &h, float &pdf) // microfacet.cpp:8-19
{
    const float a2 = sqr(alpha);
    const float tan2 = a2 * xi0 / (1.0f - xi0), phi = k2Pi * xi1;
    const float cos_t = static_cast<float>(1.0 / sqrt(D(1.0f + tan2)));
    const float sin_t = sqrtf(1.0f - sqr(cos_t));
    h = V3{sin_t * gl::cosf(phi), sin_t * gl::sinf(phi), cos_t};
    pdf = static_cast<float>(1.0 / (D(kPi * a2) * pow3d(cos_t) * D(sqr(1.0f + tan2 / a2))));
}

MCPT_HD float smith_g1_iso(float alpha, V3 v, V3 h) // microfacet.cpp:62-74
{
    const float n_dot_v = v.z;
    if (n_dot_v * h.z <= 0)
        return 0;
    const float c2 = sqr(n_dot_v), tan2 = (1.0f - c2) / c2, a2 = sqr(alpha);
    return 2.0f / (1.0f + sqrtf(static_cast<float>(1.0 + D(a2 * tan2))));
}

} // namespace mcpt

#endif // MCPT_VECMATH_H
