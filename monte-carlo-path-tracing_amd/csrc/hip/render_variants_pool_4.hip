// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h): pool walk outside LDS, the FULL feature set
// (volume paths, quadrics, every BSDF model, emitter records, textures) with and without the sliver rules — meshes that the
// surface-materials instantiations do not cover (the reference's `box` scene).
#define MCPT_UNIT_POOL_4
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kAll | kPB, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kAll | kPB | kS, false, false>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
