// Launch interface between the C-ABI layer (capi.cpp) and the HIP kernel.
#ifndef MCPT_RENDER_KERNEL_H
#define MCPT_RENDER_KERNEL_H

#include <hip/hip_runtime_api.h>

#include "../device_scene.h"

namespace mcpt
{

constexpr int kBlockSize = 256;
// Traversal data (both hierarchies + triangle positions) up to this size is staged
// in LDS, next to the traversal stacks (walk_depth x 256 x 4 B per workgroup):
// 4 workgroups per CU x (24 KiB + stacks) leaves room in the 160 KiB of a CU.
constexpr size_t kLdsGeometryBytes = 24 * 1024;

// Work description of one launch: pixels are enumerated tile by tile
// (8x8 pixel tiles, row-major tile order); item q -> local tile q / 64, pixel
// q % 64 inside it; global tile = tile_first + local_tile * tile_stride.
struct RenderJob
{
    uint32_t n_items;     // 64 * number of tiles handled by this launch
    uint32_t tile_first;  // first global tile
    uint32_t tile_stride; // distance between consecutive tiles of this launch
    uint32_t tiles_x;     // tiles per image row
    uint32_t packed;      // 0: write frame layout, 1: write packed tile layout
    uint32_t reference_walk; // 1: force the reference-order walk (validation); masks force it anyway
};

hipError_t LaunchRender(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters,
                        hipStream_t stream, uint32_t n_cus, const char **variant);

// Unit kernels for diagnostics and parity tests (one query per lane).
hipError_t LaunchIntersect(const DeviceScene &sc, uint32_t n, const float *rays, const uint32_t *seeds, float *out,
                           uint32_t *seeds_out, bool reference_walk, hipStream_t stream);
hipError_t LaunchTracePixel(const DeviceScene &sc, uint32_t pixel, uint32_t capacity, float *out, uint32_t *n_steps,
                            hipStream_t stream);
hipError_t LaunchBsdf(const DeviceScene &sc, uint32_t n, uint32_t id_bsdf, int mode, const float *recs,
                      const uint32_t *seeds, float *out, uint32_t *seeds_out, hipStream_t stream);

} // namespace mcpt

#endif // MCPT_RENDER_KERNEL_H
