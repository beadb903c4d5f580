// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h): the lane-owns-a-path kernel with the
// wavefront-cooperative pool walk (pool_walk.h) for scenes OUTSIDE LDS — 32-bit items, the 4-wide exact hierarchy read
// through the caches.
#define MCPT_UNIT_POOL
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kSurface | kPB, false, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kSurface | kPB | kS, false, false>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
