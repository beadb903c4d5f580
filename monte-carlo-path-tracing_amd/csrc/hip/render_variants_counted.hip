// Render-kernel instantiations compiled in this unit (see render_kernel_impl.h).
#define MCPT_UNIT_COUNTED
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kAll, true, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kAll | kV | kS, true, false>(MCPT_LAUNCH_ARGS);
// ... and the counting mode of the pool-walk kernels (any scene class: the hierarchy through the caches, 32-bit items)
template hipError_t Launch<kAll | kPB | kS, true, false>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kAll | kP, true, true>(MCPT_LAUNCH_ARGS); // the LDS form of the pool walk (exact 128-byte records, 16-bit items): cornell-class scenes

} // namespace mcpt
