// Render kernels of the LOW-DISCREPANCY build (hip/lowdisc_units.h): the pool walk on scenes outside LDS with the full feature
// set (volume paths, quadrics, every BSDF model, emitter records, textures), with and without the sliver rules.
#define MCPT_UNIT_LOWDISC_4
#include "lowdisc_units.h"

namespace mcpt
{

template hipError_t Launch<kAll | kPB | kLD, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kAll | kPB | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
