// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h).
#define MCPT_UNIT_SURFACE
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kSurface | kV, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kSurface | kV | kS, false, false>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
