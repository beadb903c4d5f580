// Device build of the reference-topology LBVH (lbvh_build.hip).
#ifndef MCPT_LBVH_BUILD_H
#define MCPT_LBVH_BUILD_H

#include <hip/hip_runtime_api.h>

#include "../device_scene.h"

namespace mcpt
{

// boxes_dev: 6 floats per primitive (lo.xyz, hi.xyz); areas_dev: 1 per primitive.
// nodes_dev: 2 float4 per node, node_area_dev: 1 float per node, 2n-1 nodes in the
// layout of device_scene.h (`nodes`, tree-local links, object = primitive index).
// All pointers are device memory.  Synchronous with respect to `stream`.
hipError_t BuildLbvhOnDevice(uint32_t n, const float *boxes_dev, const float *areas_dev, float4 *nodes_dev,
                             float *node_area_dev, hipStream_t stream);

} // namespace mcpt

#endif // MCPT_LBVH_BUILD_H
