// Render kernels of the LOW-DISCREPANCY build (hip/lowdisc_units.h): the lean instantiations of LDS-resident scenes —
// diffuse surfaces with the pool walk (cornell-box), the same with emitter records, volume paths without emitter / texture
// code (volumetric-caustic).
#define MCPT_UNIT_LOWDISC_2
#include "lowdisc_units.h"

namespace mcpt
{

template hipError_t Launch<kP | kLD, false, true>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kFeatEmitters | kP | kLD, false, true>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kVolumeLean | kO | kLD, false, true>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
