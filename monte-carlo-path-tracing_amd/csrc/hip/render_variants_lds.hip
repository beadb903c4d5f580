// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h): the full-feature kernel for scenes
// whose traversal data sits in LDS (volumetric-caustic: 14 triangles and a sphere in a medium).
#define MCPT_UNIT_LDS
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kAll | kO, false, true>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
