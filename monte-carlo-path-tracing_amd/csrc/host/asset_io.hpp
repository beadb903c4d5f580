// Mesh and image readers of the scene front end (see asset_io.cpp).
#ifndef MCPT_HOST_ASSET_IO_HPP
#define MCPT_HOST_ASSET_IO_HPP

#include <cstdint>
#include <string>
#include <vector>

namespace mcpt
{

struct MeshData
{
    std::vector<float> positions; // 3 per vertex
    std::vector<float> normals;   // 3 per vertex or empty
    std::vector<float> texcoords; // 2 per vertex or empty
    std::vector<uint32_t> indices; // 3 per triangle
};

struct ImageData
{
    int width = 0, height = 0, channel = 0;
    std::vector<float> data; // row 0 = top
};

MeshData LoadObj(const std::string &path, bool flip_texcoords, bool face_normals);
MeshData LoadSerialized(const std::string &path, int shape_index);
ImageData LoadFloatImage(const std::string &path);

// One PIZ-compressed EXR block -> raw scanline bytes (exr_piz.cpp).  `words_per_sample`: per
// channel in file order, 1 for HALF and 2 for FLOAT / UINT.
void DecodePizBlock(const uint8_t *src, size_t n_src, const std::vector<int> &words_per_sample, int width, int lines,
                    uint8_t *dst);

} // namespace mcpt

#endif // MCPT_HOST_ASSET_IO_HPP
