// Texture evaluation (constant / checkerboard / bilinear bitmap), shared by
// the host commit (env-map table build) and the device shading code.
// Follows reference src/renderer/textures/{texture,bitmap,checkboard,
// constant_texture}.cpp.
#ifndef MCPT_TEXTURES_H
#define MCPT_TEXTURES_H

#include "vecmath.h"

namespace mcpt
{

struct BitmapTap
{
    uint32_t x0, y0, x1, y1;
    float tx, ty;
};

// bitmap.cpp:8-24: wrap by repeated add/subtract, truncate, pick neighbours.
MCPT_HD BitmapTap bitmap_tap(const TextureRec &t, V2 uv)
{
    const V3 p = transform_point(t.to_uv, V3{uv.u, uv.v, 0.0f});
    float x = p.x * t.width, y = p.y * t.height;
    // Beyond 2^24 texture widths a subtraction of the width no longer changes x and the reference's loop
    // never ends (a NaN leaves the loops and is then cast to an index): such coordinates read texel 0
    // here instead of hanging a persistent kernel.  Below that the repeated arithmetic is the reference's.
    if (!(fabsf(x) < 16777216.0f * t.width))
        x = 0.0f;
    if (!(fabsf(y) < 16777216.0f * t.height))
        y = 0.0f;
    while (x < 0)
        x += t.width;
    while (x > t.width - 1)
        x -= t.width;
    while (y < 0)
        y += t.height;
    while (y > t.height - 1)
        y -= t.height;
    BitmapTap tap;
    tap.x0 = static_cast<uint32_t>(x), tap.y0 = static_cast<uint32_t>(y);
    tap.tx = x - tap.x0, tap.ty = y - tap.y0;
    tap.x1 = (tap.tx > 0.0f) ? tap.x0 + 1 : tap.x0;
    tap.y1 = (tap.ty > 0.0f) ? tap.y0 + 1 : tap.y0;
    return tap;
}

MCPT_HD V3 bitmap_color(const TextureRec &t, const float *texels, V2 uv) // bitmap.cpp:6-55
{
    const BitmapTap k = bitmap_tap(t, uv);
    const float *px = texels + t.texel_base;
    const uint32_t w = static_cast<uint32_t>(t.width);
    if (t.channel == 1)
    {
        const float c00 = px[k.x0 + w * k.y0], c01 = px[k.x0 + w * k.y1],
                    c10 = px[k.x1 + w * k.y0], c11 = px[k.x1 + w * k.y1];
        return splat(lerp(lerp(c00, c01, k.ty), lerp(c10, c11, k.ty), k.tx));
    }
    const uint32_t c = static_cast<uint32_t>(t.channel);
    const uint32_t o00 = (k.x0 + w * k.y0) * c, o01 = (k.x0 + w * k.y1) * c,
                   o10 = (k.x1 + w * k.y0) * c, o11 = (k.x1 + w * k.y1) * c;
    const V3 c00 = V3{px[o00], px[o00 + 1], px[o00 + 2]}, c01 = V3{px[o01], px[o01 + 1], px[o01 + 2]},
             c10 = V3{px[o10], px[o10 + 1], px[o10 + 2]}, c11 = V3{px[o11], px[o11 + 1], px[o11 + 2]};
    return lerp(lerp(c00, c01, k.ty), lerp(c10, c11, k.ty), k.tx);
}

MCPT_HD V3 checker_color(const TextureRec &t, V2 uv) // checkboard.cpp:6-21
{
    V3 p = transform_point(t.to_uv, V3{uv.u, uv.v, 0.0f});
    if (!(fabsf(p.x) < 16777216.0f)) // see bitmap_tap
        p.x = 0.0f;
    if (!(fabsf(p.y) < 16777216.0f))
        p.y = 0.0f;
    while (p.x > 1)
        p.x -= 1;
    while (p.x < 0)
        p.x += 1;
    while (p.y > 1)
        p.y -= 1;
    while (p.y < 0)
        p.y += 1;
    const int x = 2 * (static_cast<int>(p.x * 2) % 2) - 1, y = 2 * (static_cast<int>(p.y * 2) % 2) - 1;
    return (x * y == 1) ? from(t.color0) : from(t.color1);
}

// `all_constant`: the caller knows (scene feature bits) that every texture of the
// scene is a constant; passed as a compile-time constant from the kernel
// instantiations for such scenes so that the checkerboard / bitmap code (with its
// wrap-around loops) is not compiled into them at every call site.
MCPT_COLD V3 texture_color_varying(const TextureRec *textures, const float *texels, uint32_t id, V2 uv)
{
    const TextureRec &t = textures[id];
    if (t.kind == kTexChecker)
        return checker_color(t, uv);
    return bitmap_color(t, texels, uv);
}
MCPT_HD V3 texture_color(const TextureRec *textures, const float *texels, uint32_t id, V2 uv,
                         bool all_constant = false) // texture.cpp:63-78
{
    const TextureRec &t = textures[id];
    if (all_constant || t.kind == kTexConstant)
        return from(t.color);
    return texture_color_varying(textures, texels, id, uv);
}

MCPT_HD V2 texture_gradient(const TextureRec *textures, const float *texels, uint32_t id, V2 uv) // texture.cpp:80-95
{
    if (textures[id].kind == kTexConstant)
        return V2{0.0f, 0.0f};
    constexpr float delta = 1e-4f, norm = 1.0f / delta; // bitmap.cpp:57-68, checkboard.cpp:23-33
    const float v = length(texture_color(textures, texels, id, uv)),
                vu = length(texture_color(textures, texels, id, uv + V2{delta, 0})),
                vv = length(texture_color(textures, texels, id, uv + V2{0, delta}));
    return V2{(vu - v) * norm, (vv - v) * norm};
}

// texture.cpp:97-113: stochastic opacity; one draw for a constant texture and
// for a 4-channel bitmap, none otherwise.
MCPT_HD bool texture_transparent(const TextureRec *textures, const float *texels, uint32_t id, V2 uv, uint32_t &rng)
{
    const TextureRec &t = textures[id];
    if (t.kind == kTexConstant)
        return t.color.x < lcg_next(rng);
    if (t.kind != kTexBitmap || t.channel != 4)
        return false;
    const BitmapTap k = bitmap_tap(t, uv);
    const float *px = texels + t.texel_base;
    const uint32_t w = static_cast<uint32_t>(t.width);
    const float c00 = px[(k.x0 + w * k.y0) * 4 + 3], c01 = px[(k.x0 + w * k.y1) * 4 + 3],
                c10 = px[(k.x1 + w * k.y0) * 4 + 3], c11 = px[(k.x1 + w * k.y1) * 4 + 3];
    return lerp(lerp(c00, c01, k.ty), lerp(c10, c11, k.ty), k.tx) < lcg_next(rng);
}

MCPT_HD float luminance(V3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; } // envmap.cpp:9-12

} // namespace mcpt

#endif // MCPT_TEXTURES_H
