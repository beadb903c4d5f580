// TEST INFRASTRUCTURE — CPU oracle, part 1: arithmetic vocabulary.
//
// Scalar restatement of the reference's value types and numeric helpers.
// Rounding-relevant conventions that are reproduced on purpose:
//   * vector / scalar division multiplies by the reciprocal
//     (reference src/tensor/vec3.cpp:107-112,147-151, vec4.cpp, vec2.cpp);
//   * normalisation is v * (1 / |v|) (vec3.cpp:181-185);
//   * direction transforms normalise their result (vec4.hpp:55, mat4.cpp:270);
//   * several helpers call the *double* libm functions (sqrt/pow/log/exp/fabs)
//     on float arguments because only <cmath> is in scope — the
//     promotions are written out explicitly below.
// Compile with -ffp-contract=off; no fast-math.
#ifndef ORACLE_MATH_HPP
#define ORACLE_MATH_HPP

#include <cmath>
#include <cstdint>
#include <limits>

namespace orc
{

constexpr uint32_t kNone = 0xFFFFFFFFu;               // defs.hpp:22
constexpr float kEpsFloat = std::numeric_limits<float>::epsilon(); // defs.hpp:24
constexpr float kEpsDistance = 1e-4f;                 // defs.hpp:25
constexpr float kEps = 0.01f;                         // defs.hpp:26
constexpr float kMaxF = std::numeric_limits<float>::max();
constexpr float kLowestF = std::numeric_limits<float>::lowest();

constexpr float kPi = 3.141592653589793f;             // math.hpp:17-23
constexpr float k2Pi = 3.141592653589793f * 2.0f;
constexpr float kPiDiv2 = 3.141592653589793f * 0.5f;
constexpr float kPiDiv4 = 3.141592653589793f * 0.25f;
constexpr float k1DivPi = 1.0f / kPi;
constexpr float k1Div2Pi = 1.0f / k2Pi;
constexpr float k1Div4Pi = 1.0f / (4.0f * kPi);

inline float Radians(float degree) // math.hpp:25-28
{
    return degree * 0.01745329251994329576923690768489f;
}

struct V2
{
    float u = 0, v = 0;
};
inline V2 operator+(V2 a, V2 b) { return {a.u + b.u, a.v + b.v}; }
inline V2 operator-(V2 a, V2 b) { return {a.u - b.u, a.v - b.v}; }
inline V2 operator*(float t, V2 a) { return {t * a.u, t * a.v}; }

struct V3
{
    float x = 0, y = 0, z = 0;
    V3() = default;
    V3(float a) : x(a), y(a), z(a) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) // vec3.cpp:141-145
{
    const float k0 = 1.0f / b.x, k1 = 1.0f / b.y, k2 = 1.0f / b.z;
    return {a.x * k0, a.y * k1, a.z * k2};
}
inline V3 operator+(V3 a, float t) { return {a.x + t, a.y + t, a.z + t}; }
inline V3 operator-(V3 a, float t) { return {a.x - t, a.y - t, a.z - t}; }
inline V3 operator*(V3 a, float t) { return {a.x * t, a.y * t, a.z * t}; }
inline V3 operator/(V3 a, float t) // vec3.cpp:162-166
{
    const float k = 1.0f / t;
    return {a.x * k, a.y * k, a.z * k};
}
inline V3 operator+(float t, V3 a) { return {t + a.x, t + a.y, t + a.z}; }
inline V3 operator-(float t, V3 a) { return {t - a.x, t - a.y, t - a.z}; }
inline V3 operator*(float t, V3 a) { return {t * a.x, t * a.y, t * a.z}; }
inline V3 &operator+=(V3 &a, V3 b) { return a = a + b; }
inline V3 &operator*=(V3 &a, V3 b) { return a = a * b; }
inline V3 &operator*=(V3 &a, float t) { return a = a * t; }

inline float Dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float Len(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 Unit(V3 a) // vec3.cpp:181-185
{
    const float k = 1.0f / Len(a);
    return a * k;
}
inline V3 Cross(V3 a, V3 b) // vec3.cpp:192-196 (note the -x*z + z*x form)
{
    return {a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x,
            a.x * b.y - a.y * b.x};
}
inline V3 Min3(V3 a, V3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
inline V3 Max3(V3 a, V3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
inline V3 Sqrt3(V3 a) { return {sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
inline float Sq(float t) { return t * t; }
inline V3 Sq(V3 t) { return t * t; }
inline float MaxComp(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

inline float Mix(float a, float b, float t) { return (1.0f - t) * a + t * b; } // math.hpp:74-77
inline V3 Mix(V3 a, V3 b, float t) { return (1.0f - t) * a + t * b; }
inline V3 Bary(const V3 *v, float a, float b, float c) // math.hpp:80-84
{
    return a * v[0] + b * v[1] + c * v[2];
}
inline V2 Bary(const V2 *v, float a, float b, float c)
{
    return a * v[0] + b * v[1] + c * v[2];
}

// Row-major 4x4 (tensor/mat4.hpp).  m[r][c].
struct M4
{
    float m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
};

inline M4 FromRowMajor(const float *p)
{
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = p[4 * i + j];
    return r;
}

inline float Dot4(const float *a, const float *b) // vec4.cpp:164-167
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

inline M4 Transpose(const M4 &a)
{
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[j][i];
    return r;
}

// mat4.cpp:196-199 -> Mul(Vec4, Mat4) (mat4.cpp:206-211): row · column.
inline M4 MatMul(const M4 &a, const M4 &b)
{
    const M4 bt = Transpose(b);
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = Dot4(a.m[i], bt.m[j]);
    return r;
}

inline M4 Translation(V3 t) // mat4.cpp:229-235
{
    M4 r;
    r.m[0][3] = t.x, r.m[1][3] = t.y, r.m[2][3] = t.z;
    return r;
}

// Cofactor-expansion inverse, same term grouping as mat4.cpp:110-168
// (which follows the glm formulation): six 2x2 "coef" families, four signed
// cofactor columns, determinant from the first row.
inline M4 Inverse(const M4 &a)
{
    const float(*r)[4] = a.m;
    // 2x2 sub-determinants of rows 1..3, named by the column pair (p,q) and
    // the row pair: c{pq}_{rows}
    auto sub = [&](int ra, int rb, int ca, int cb)
    { return r[ra][ca] * r[rb][cb] - r[rb][ca] * r[ra][cb]; };
    const float c00 = sub(2, 3, 2, 3), c02 = sub(1, 3, 2, 3), c03 = sub(1, 2, 2, 3);
    const float c04 = sub(2, 3, 1, 3), c06 = sub(1, 3, 1, 3), c07 = sub(1, 2, 1, 3);
    const float c08 = sub(2, 3, 1, 2), c10 = sub(1, 3, 1, 2), c11 = sub(1, 2, 1, 2);
    const float c12 = sub(2, 3, 0, 3), c14 = sub(1, 3, 0, 3), c15 = sub(1, 2, 0, 3);
    const float c16 = sub(2, 3, 0, 2), c18 = sub(1, 3, 0, 2), c19 = sub(1, 2, 0, 2);
    const float c20 = sub(2, 3, 0, 1), c22 = sub(1, 3, 0, 1), c23 = sub(1, 2, 0, 1);

    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07},
                f2[4] = {c08, c08, c10, c11}, f3[4] = {c12, c12, c14, c15},
                f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {r[1][0], r[0][0], r[0][0], r[0][0]},
                v1[4] = {r[1][1], r[0][1], r[0][1], r[0][1]},
                v2[4] = {r[1][2], r[0][2], r[0][2], r[0][2]},
                v3[4] = {r[1][3], r[0][3], r[0][3], r[0][3]};
    const float sa[4] = {+1.0f, -1.0f, +1.0f, -1.0f},
                sb[4] = {-1.0f, +1.0f, -1.0f, +1.0f};
    float inv[4][4];
    for (int k = 0; k < 4; ++k)
    {
        inv[0][k] = (v1[k] * f0[k] - v2[k] * f1[k] + v3[k] * f2[k]) * sa[k];
        inv[1][k] = (v0[k] * f0[k] - v2[k] * f3[k] + v3[k] * f4[k]) * sb[k];
        inv[2][k] = (v0[k] * f1[k] - v1[k] * f3[k] + v3[k] * f5[k]) * sa[k];
        inv[3][k] = (v0[k] * f2[k] - v1[k] * f4[k] + v2[k] * f5[k]) * sb[k];
    }
    const float d0 = r[0][0] * inv[0][0], d1 = r[0][1] * inv[1][0],
                d2 = r[0][2] * inv[2][0], d3 = r[0][3] * inv[3][0];
    const float det = (d0 + d1) + (d2 + d3);
    const float rcp = 1.0f / det;
    M4 out;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            out.m[i][j] = rcp * inv[i][j];
    return out;
}

inline V3 XformPoint(const M4 &a, V3 p) // mat4.cpp:264-267, vec4.cpp:93-97
{
    const float v[4] = {p.x, p.y, p.z, 1.0f};
    const float x = Dot4(a.m[0], v), y = Dot4(a.m[1], v), z = Dot4(a.m[2], v),
                w = Dot4(a.m[3], v);
    const float k = 1.0f / w;
    return {x * k, y * k, z * k};
}

inline V3 XformDir(const M4 &a, V3 d) // mat4.cpp:270-273, vec4.hpp:55
{
    const float v[4] = {d.x, d.y, d.z, 0.0f};
    return Unit({Dot4(a.m[0], v), Dot4(a.m[1], v), Dot4(a.m[2], v)});
}

// ---- random numbers and low-discrepancy points (math.hpp:29-63) ----------

inline uint32_t Tea4(uint32_t v0, uint32_t v1) // math.hpp:43-54, 4 rounds
{
    uint32_t sum = 0;
    for (int round = 0; round < 4; ++round)
    {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

inline float Rand(uint32_t *state) // math.hpp:58-63: LCG, top 24 bits unused
{
    *state = *state * 1664525u + 1013904223u;
    return static_cast<float>(*state & 0x00ffffffu) /
           static_cast<float>(0x01000000u);
}

// math.hpp:29-41.  The index is updated through a float multiply
// (`index *= base_inv`), i.e. uint -> float, multiply, truncate back.
template <uint32_t base>
inline float RadicalInverse(uint32_t index)
{
    const float base_inv = 1.0f / base;
    float result = 0.0f, frac = base_inv;
    while (index > 0)
    {
        result += frac * (index % base);
        index = static_cast<uint32_t>(static_cast<float>(index) * base_inv);
        frac *= base_inv;
    }
    return result;
}

inline float PowerHeuristic(float a, float b) // math.cpp:8-13
{
    a *= a;
    b *= b;
    return a / (a + b);
}

inline V3 ConeUniform(float cos_cutoff, float xi0, float xi1) // math.cpp:15-22
{
    const float cos_t = 1.0f - (1.0f - cos_cutoff) * xi0, phi = 2.0f * kPi * xi1;
    const float sin_t = static_cast<float>(
        sqrt(static_cast<double>(fmaxf(0.0f, 1.0f - cos_t * cos_t))));
    return {sin_t * cosf(phi), sin_t * sinf(phi), cos_t};
}

inline V3 SphereUniform(float xi0, float xi1) // math.cpp:24-29
{
    const float cos_t = 1.0f - 2.0f * xi0, phi = k2Pi * xi1;
    const float sin_t = sqrtf(1.0f - Sq(cos_t));
    return {sin_t * cosf(phi), sin_t * sinf(phi), cos_t};
}

inline void HemisphereCosine(float xi0, float xi1, V3 *dir, float *pdf) // math.cpp:31-38
{
    const float cos_t = sqrtf(xi0), phi = k2Pi * xi1;
    const float sin_t =
        static_cast<float>(sqrt(static_cast<double>(1.0f - Sq(cos_t))));
    *dir = {sin_t * cosf(phi), sin_t * sinf(phi), cos_t};
    *pdf = k1DivPi * cos_t;
}

inline uint32_t CdfSearch(uint32_t num, const float *cdf, float target) // math.cpp:40-55
{
    uint32_t lo = 0, hi = num;
    while (lo + 1 != hi)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] < target)
            lo = mid;
        else if (cdf[mid] > target)
            hi = mid;
        else
            return mid;
    }
    return hi;
}

inline bool Quadratic(float a, float b, float c, float *x0, float *x1) // math.cpp:57-98
{
    if (a == 0.0f)
    {
        if (b != 0.0f)
        {
            *x0 = *x1 = -c / b;
            return true;
        }
        return false;
    }
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f)
        return false;
    const float root = sqrtf(disc);
    const float q = (b < 0.0f) ? -0.5f * (b - root) : -0.5f * (b + root);
    *x0 = q / a;
    *x1 = c / q;
    if (*x0 > *x1)
    {
        const float t = *x0;
        *x0 = *x1;
        *x1 = t;
    }
    return true;
}

// y-up spherical coordinates (math.cpp:101-128).
inline void ToSpherical(V3 v, float *theta, float *phi, float *r)
{
    if (r != nullptr)
        *r = Len(v);
    v = Unit(v);
    *theta = acosf(fminf(1.0f, fmaxf(-1.0f, v.y)));
    if (v.z == 0 && v.x == 0)
    {
        *phi = 0;
    }
    else
    {
        *phi = atan2f(v.z, v.x);
        if (*phi < 0.0f)
            *phi += 2.0f * kPi;
    }
}

inline V3 FromSpherical(float theta, float phi, float r)
{
    const float sin_t = sinf(theta);
    return {r * sinf(phi) * sin_t, r * cosf(theta), r * cosf(phi) * sin_t};
}

// math.cpp:130-145.  1/sqrt(...) is a float divided by a double sqrt.
inline V3 FrameToWorld(V3 local, V3 up)
{
    V3 c;
    if (sqrt(static_cast<double>(Sq(up.x) + Sq(up.z))) > kEpsFloat)
    {
        const float k = static_cast<float>(
            1.0f / sqrt(static_cast<double>(Sq(up.x) + Sq(up.z))));
        c = {up.z * k, 0, -up.x * k};
    }
    else
    {
        const float k = static_cast<float>(
            1.0f / sqrt(static_cast<double>(Sq(up.y) + Sq(up.z))));
        c = {0, up.z * k, -up.y * k};
    }
    const V3 b = Unit(Cross(c, up));
    return Unit(local.x * b + local.y * c + local.z * up);
}

// math.cpp:148-166 (matrix flavour, used for cylinders).
inline M4 FrameMatrix(V3 up)
{
    V3 c;
    if (sqrt(static_cast<double>(Sq(up.x) + Sq(up.z))) > kEpsFloat)
    {
        const float k = static_cast<float>(
            1.0f / sqrt(static_cast<double>(Sq(up.x) + Sq(up.z))));
        c = {-up.z * k, 0, up.x * k};
    }
    else
    {
        const float k = static_cast<float>(
            1.0f / sqrt(static_cast<double>(Sq(up.y) + Sq(up.z))));
        c = {0, -up.z * k, up.y * k};
    }
    const V3 b = Unit(Cross(c, up));
    M4 r;
    r.m[0][0] = b.x, r.m[0][1] = b.y, r.m[0][2] = b.z, r.m[0][3] = 0;
    r.m[1][0] = c.x, r.m[1][1] = c.y, r.m[1][2] = c.z, r.m[1][3] = 0;
    r.m[2][0] = up.x, r.m[2][1] = up.y, r.m[2][2] = up.z, r.m[2][3] = 0;
    r.m[3][0] = 0, r.m[3][1] = 0, r.m[3][2] = 0, r.m[3][3] = 1;
    return r;
}

inline V3 Reflect(V3 wi, V3 n) // ray.cpp:49-52
{
    return Unit(wi - 2.0f * Dot(wi, n) * n);
}

inline bool Refract(V3 wi, V3 n, float eta_inv, V3 *wt) // ray.cpp:54-68
{
    const float cos_t = static_cast<float>(fabs(static_cast<double>(Dot(wi, n))));
    const float k = 1.0f - Sq(eta_inv) * (1.0f - Sq(cos_t));
    if (k < 0)
        return false;
    *wt = Unit(eta_inv * wi + (eta_inv * cos_t - sqrtf(k)) * n);
    return true;
}

} // namespace orc

#endif // ORACLE_MATH_HPP
