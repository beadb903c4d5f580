// Host-only 4x4 matrix helpers with the reference's arithmetic
// (src/tensor/mat4.cpp): shared by the scene commit and the XML front end.
#ifndef MCPT_HOST_MATRIX_HPP
#define MCPT_HOST_MATRIX_HPP

#include <cstring>

#include "../vecmath.h"

namespace mcpt
{

// ---- host-only 4x4 helpers (reference src/tensor/mat4.cpp) -----------------
inline Mat4f Identity()
{
    Mat4f r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}

inline Mat4f Load(const float *p)
{
    Mat4f r;
    std::memcpy(r.m, p, sizeof(r.m));
    return r;
}

inline Mat4f Transposed(const Mat4f &a)
{
    Mat4f r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[4 * i + j] = a.m[4 * j + i];
    return r;
}

inline Mat4f Multiply(const Mat4f &a, const Mat4f &b) // mat4.cpp:196-211: row . column, left to right
{
    Mat4f r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[4 * i + j] = a.m[4 * i + 0] * b.m[0 + j] + a.m[4 * i + 1] * b.m[4 + j] +
                             a.m[4 * i + 2] * b.m[8 + j] + a.m[4 * i + 3] * b.m[12 + j];
    return r;
}

// mat4.cpp:110-168 — cofactor inverse in the reference's term grouping: 18
// 2x2 minors of rows 1..3, four cofactor columns built as a*b - c*d + e*f,
// alternating signs, determinant from row 0 as (p0 + p1) + (p2 + p3).
inline Mat4f Inverted(const Mat4f &a)
{
    const float *r0 = a.m, *r1 = a.m + 4, *r2 = a.m + 8, *r3 = a.m + 12;
    enum { X, Y, Z, W };
    auto minor2 = [](const float *p, const float *q, int c0, int c1)
    { return p[c0] * q[c1] - q[c0] * p[c1]; };
    // for each column pair, the minors over row pairs (2,3), (1,3), (1,2)
    const float zw[3] = {minor2(r2, r3, Z, W), minor2(r1, r3, Z, W), minor2(r1, r2, Z, W)};
    const float yw[3] = {minor2(r2, r3, Y, W), minor2(r1, r3, Y, W), minor2(r1, r2, Y, W)};
    const float yz[3] = {minor2(r2, r3, Y, Z), minor2(r1, r3, Y, Z), minor2(r1, r2, Y, Z)};
    const float xw[3] = {minor2(r2, r3, X, W), minor2(r1, r3, X, W), minor2(r1, r2, X, W)};
    const float xz[3] = {minor2(r2, r3, X, Z), minor2(r1, r3, X, Z), minor2(r1, r2, X, Z)};
    const float xy[3] = {minor2(r2, r3, X, Y), minor2(r1, r3, X, Y), minor2(r1, r2, X, Y)};
    // lane k of the reference's fac vectors uses minor index {0,0,1,2}[k];
    // lane k of its vec vectors uses row 1 for k = 0 and row 0 otherwise.
    static const int pick[4] = {0, 0, 1, 2};
    float adj[4][4];
    for (int k = 0; k < 4; ++k)
    {
        const float *row = (k == 0) ? r1 : r0;
        const int s = pick[k];
        const float sign_a = (k % 2 == 0) ? 1.0f : -1.0f, sign_b = -sign_a;
        adj[0][k] = (row[Y] * zw[s] - row[Z] * yw[s] + row[W] * yz[s]) * sign_a;
        adj[1][k] = (row[X] * zw[s] - row[Z] * xw[s] + row[W] * xz[s]) * sign_b;
        adj[2][k] = (row[X] * yw[s] - row[Y] * xw[s] + row[W] * xy[s]) * sign_a;
        adj[3][k] = (row[X] * yz[s] - row[Y] * xz[s] + row[Z] * xy[s]) * sign_b;
    }
    const float p0 = r0[X] * adj[0][0], p1 = r0[Y] * adj[1][0], p2 = r0[Z] * adj[2][0],
                p3 = r0[W] * adj[3][0];
    const float rcp_det = 1.0f / ((p0 + p1) + (p2 + p3));
    Mat4f out;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            out.m[4 * i + j] = rcp_det * adj[i][j];
    return out;
}

inline Mat4f TranslationMatrix(V3 t) // mat4.cpp:229-235
{
    Mat4f r = Identity();
    r.m[3] = t.x, r.m[7] = t.y, r.m[11] = t.z;
    return r;
}

} // namespace mcpt

#endif // MCPT_HOST_MATRIX_HPP
