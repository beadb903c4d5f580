// Instantiations of the lane-owns-a-path kernel: the lean LDS-resident pool-walk kernels with two queries per vertex (cornell's
// class: the headline workload's kernel).
#define MCPT_UNIT_LEAN_POOL
#include "render_kernel_impl.h"

namespace mcpt
{

template hipError_t Launch<kP, false, true>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kFeatEmitters | kP, false, true>(MCPT_LAUNCH_ARGS);

} // namespace mcpt
