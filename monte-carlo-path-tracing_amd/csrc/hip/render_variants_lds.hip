// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h): the full-feature kernel for scenes
// whose traversal data sits in LDS (volumetric-caustic: 14 triangles and a sphere in a medium).
#define MCPT_UNIT_LDS
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kAll | kO, false, true>(MCPT_LAUNCH_ARGS);
// ... and the same without the emitter and texture code, for scenes that have neither (volumetric-caustic: an area
// light and constant textures only): 6 spilled VGPRs instead of 152 at the same 168-register budget
template hipError_t Launch<kVolumeLean | kO, false, true>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
