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
    std::vector<float> tangents, bitangents; // 3 per vertex or empty (mesh_postprocess.cpp)
    std::vector<uint32_t> indices; // 3 per triangle
};

// The importer post-processing the reference requests for OBJ / PLY (mesh_postprocess.cpp).
void GenerateSmoothNormals(MeshData &m);
void CalcTangentSpace(MeshData &m);

struct ImageData
{
    int width = 0, height = 0, channel = 0;
    std::vector<float> data; // row 0 = top
};

// How the reference treats the `gamma` of a bitmap depends on the file type
// (src/utils/image_io.cpp:75-147).
enum class ImageKind
{
    kExr,  // exponent applied to the first width*height floats only (reference quirk), none when 0
    kHdr,  // -1: sRGB decode, otherwise exponent, none when 0
    kLdr,  // 8-bit: 0 or -1: sRGB decode, otherwise exponent
    kPfm,  // (not read by the reference) exponent, none when 0
};

MeshData LoadObj(const std::string &path, bool flip_texcoords, bool face_normals);
MeshData LoadSerialized(const std::string &path, int shape_index);
MeshData LoadPly(const std::string &path, bool face_normals);
// glTF 2.0 (.gltf / .glb; gltf_io.cpp): the reference's assimp import + ProcessAssimpNode restated (node transforms ignored, like there)
MeshData LoadGltf(const std::string &path, bool face_normals);
// `gamma` is applied per the file type's rule; the result is what the reference's
// image_io::Read returns (before its optional down-scaling).
ImageData LoadFloatImage(const std::string &path, float gamma = 0.0f);

// stb_image_resize2's default float down-scaling (image_resize.cpp): what the reference applies
// to an environment map wider than the film can resolve.
ImageData ResizeLikeReference(const ImageData &in, int out_width, int out_height);
void LoadJpeg8(const std::string &path, int &width, int &height, int &channel, std::vector<uint8_t> &pixels);
void LoadPng8(const std::string &path, int &width, int &height, int &channel, std::vector<uint8_t> &pixels);
ImageData LoadRadianceHdr(const std::string &path);

// One PIZ-compressed EXR block -> raw scanline bytes (exr_piz.cpp).  `words_per_sample`: per
// channel in file order, 1 for HALF and 2 for FLOAT / UINT.
void DecodePizBlock(const uint8_t *src, size_t n_src, const std::vector<int> &words_per_sample, int width, int lines,
                    uint8_t *dst);

} // namespace mcpt

#endif // MCPT_HOST_ASSET_IO_HPP
