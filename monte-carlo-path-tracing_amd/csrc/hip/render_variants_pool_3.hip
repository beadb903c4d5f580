// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h): pool walk outside LDS, surface materials
// with ONE BSDF model beyond diffuse compiled in (matpreview rough conductor / rough dielectric: fewer spilled registers).
#define MCPT_UNIT_POOL_3
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kSurface | kPBU | kFeatConductorOnly, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kSurface | kPBU | kFeatDielectricOnly, false, false>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
