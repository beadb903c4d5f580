/* TEST INFRASTRUCTURE.  The image decoder and resizer the reference uses, compiled from the
 * headers it vendors (REFERENCE/extern/stb/stb_image.h, stb_image_resize2.h) where they lie —
 * nothing of them is copied here.  The reference calls them in src/utils/image_io.cpp:100-183
 * (stbi_load / stbi_loadf for PNG, JPEG, HDR ...; stbir_resize_float_linear for environment
 * maps wider than the film resolves).  The product's own readers (csrc/host/jpeg_io.cpp,
 * ldr_hdr_io.cpp, image_resize.cpp) are pinned against this library by
 * tests/test_xml_frontend.py and by the golden vectors tests/golden/stb_vectors.npz that
 * tests/golden/make_stb_golden.py generates with it.  Built by `make ref` into
 * oracle/_ref/libstb_ref.so. */
#define STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_RESIZE_IMPLEMENTATION
#include "stb_image.h"
#include "stb_image_resize2.h"

#include <string.h>

/* 8-bit decode, channels as in the file.  out: capacity bytes.  Returns 0, -1 (decode failed),
 * -2 (capacity too small; dims are set). */
int mcpt_stb_load8(const char *path, int *w, int *h, int *c, unsigned char *out, size_t capacity)
{
    unsigned char *p = stbi_load(path, w, h, c, 0);
    if (!p)
        return -1;
    const size_t n = (size_t)*w * *h * *c;
    int rc = 0;
    if (n > capacity)
        rc = -2;
    else
        memcpy(out, p, n);
    stbi_image_free(p);
    return rc;
}

/* Float decode (Radiance .hdr natively; LDR files through stb's ldr-to-hdr conversion). */
int mcpt_stb_loadf(const char *path, int *w, int *h, int *c, float *out, size_t capacity)
{
    float *p = stbi_loadf(path, w, h, c, 0);
    if (!p)
        return -1;
    const size_t n = (size_t)*w * *h * *c;
    int rc = 0;
    if (n > capacity)
        rc = -2;
    else
        memcpy(out, p, n * sizeof(float));
    stbi_image_free(p);
    return rc;
}

/* image_io::Resize (image_io.cpp:173-183). */
int mcpt_stb_resize(const float *in, int in_w, int in_h, float *out, int out_w, int out_h, int channels)
{
    return stbir_resize_float_linear(in, in_w, in_h, 0, out, out_w, out_h, 0, (stbir_pixel_layout)channels) ? 0 : -1;
}
