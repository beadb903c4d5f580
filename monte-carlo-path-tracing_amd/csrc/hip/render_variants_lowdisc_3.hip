// Render kernels of the LOW-DISCREPANCY build (hip/lowdisc_units.h): the pool walk on scenes outside LDS — diffuse surfaces
// (dragon/scene.xml) and surface materials (matpreview), with and without the sliver rules.
#define MCPT_UNIT_LOWDISC_3
#include "lowdisc_units.h"

namespace mcpt
{

template hipError_t Launch<kFeatEmitters | kPB | kLD, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kFeatEmitters | kPB | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kSurface | kPB | kLD, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kSurface | kPB | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
