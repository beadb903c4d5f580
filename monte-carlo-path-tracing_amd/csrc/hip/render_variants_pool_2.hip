// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h): pool walk outside LDS, the LEAN feature set
// (diffuse surfaces, area lights and emitter records; dragon/scene.xml) with and without the sliver rules.
#define MCPT_UNIT_POOL_2
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kFeatEmitters | kPB, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kFeatEmitters | kPB | kS, false, false>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
