// Down-scaling of an environment map that is wider than the film can resolve.
//
// The reference limits an `envmap` emitter's bitmap to width_target = film width * 360 / fov_x
// texels (src/parser/parser.cpp:1403-1409) and, when the file is wider, resizes it inside
// image_io::Read (src/utils/image_io.cpp:160-183) with stb_image_resize2's
// stbir_resize_float_linear — i.e. that library's defaults: Mitchell-Netravali filter
// (B = C = 1/3) stretched by 1 / scale for down-sampling, edge mode "clamp", per-output-pixel
// normalisation of the weights, both axes treated separately.  This file restates that
// algorithm from the library's published description; tests/test_xml_frontend.py pins it
// against the library itself, compiled from the reference's vendored header into
// oracle/_ref (results agree to float rounding — the library's SIMD summation order is not
// reproduced).  For 4 channels the library also weights colour by alpha; environment maps
// carry alpha = 1 (asset_io.cpp, LoadExr), where that is the identity, so it is not done.
#include "asset_io.hpp"

#include <cmath>

namespace mcpt
{
namespace
{

float Mitchell(float x)
{
    if (x < 0.0f)
        x = -x;
    if (x < 1.0f)
        return (16.0f + x * x * (21.0f * x - 36.0f)) / 18.0f;
    if (x < 2.0f)
        return (32.0f + x * (-60.0f + x * (36.0f - 7.0f * x))) / 18.0f;
    return 0.0f;
}

struct Taps
{
    int first = 0;
    std::vector<float> weight; // for input texels first, first + 1, ...
};

// Weights of the input texels for every output texel of one axis.
std::vector<Taps> AxisTaps(int in_size, int out_size)
{
    const float scale = static_cast<float>(out_size) / static_cast<float>(in_size);
    const float radius = 2.0f / scale; // the filter's support in input texels
    std::vector<Taps> taps(out_size);
    for (int o = 0; o < out_size; ++o)
    {
        const float out_centre = static_cast<float>(o) + 0.5f;
        const float in_centre = out_centre / scale;
        const int lo = static_cast<int>(std::floor(in_centre - radius - 0.5f)) - 1,
                  hi = static_cast<int>(std::ceil(in_centre + radius - 0.5f)) + 1;
        std::vector<float> w(in_size, 0.0f);
        int first = in_size, last = -1;
        float total = 0.0f;
        for (int i = lo; i <= hi; ++i)
        {
            const float x = out_centre - (static_cast<float>(i) + 0.5f) * scale;
            const float c = Mitchell(x) * scale;
            if (c == 0.0f)
                continue;
            const int at = i < 0 ? 0 : (i >= in_size ? in_size - 1 : i); // edge clamp: folded onto the border texel
            w[at] += c, total += c;
            first = at < first ? at : first, last = at > last ? at : last;
        }
        Taps &t = taps[o];
        if (last < first)
            continue;
        const float norm = 1.0f / total;
        t.first = first;
        t.weight.assign(w.begin() + first, w.begin() + last + 1);
        for (float &c : t.weight)
            c *= norm;
    }
    return taps;
}

} // namespace

ImageData ResizeLikeReference(const ImageData &in, int out_width, int out_height)
{
    ImageData out;
    out.width = out_width, out.height = out_height, out.channel = in.channel;
    const int ch = in.channel;
    const std::vector<Taps> tx = AxisTaps(in.width, out_width), ty = AxisTaps(in.height, out_height);
    // rows first (fewer rows survive), then columns
    std::vector<float> rows(static_cast<size_t>(out_height) * in.width * ch, 0.0f);
    for (int y = 0; y < out_height; ++y)
    {
        float *dst = &rows[static_cast<size_t>(y) * in.width * ch];
        const Taps &t = ty[y];
        for (size_t k = 0; k < t.weight.size(); ++k)
        {
            const float w = t.weight[k];
            const float *src = &in.data[static_cast<size_t>(t.first + k) * in.width * ch];
            for (int i = 0; i < in.width * ch; ++i)
                dst[i] += w * src[i];
        }
    }
    out.data.assign(static_cast<size_t>(out_width) * out_height * ch, 0.0f);
    for (int y = 0; y < out_height; ++y)
        for (int x = 0; x < out_width; ++x)
        {
            const Taps &t = tx[x];
            float *dst = &out.data[(static_cast<size_t>(y) * out_width + x) * ch];
            for (size_t k = 0; k < t.weight.size(); ++k)
            {
                const float *src = &rows[(static_cast<size_t>(y) * in.width + t.first + k) * ch];
                for (int c = 0; c < ch; ++c)
                    dst[c] += t.weight[k] * src[c];
            }
        }
    return out;
}

} // namespace mcpt
