// Instantiations of the lane-owns-a-path kernel: the lean LDS-resident pool-walk kernels with MERGED QUERIES
// (mcpt_renderer_set_pool_walk(r, 2)).
//
// This unit is compiled with -fno-slp-vectorize (csrc/Makefile), and that is a CORRECTNESS requirement with ROCm 7.2's compiler:
// with the SLP vectoriser on, gfx950's back end generates wrong code for Launch<kPM, false, true> — cornell renders with 80 % of its
// pixels darker — although the source is right (EXPERIMENTS.md R6-1: the lockstep host build of this very body is exact under every
// lane order and with poisoned pool areas; the kernel is exact at -O1, at -O3 without the SLP vectoriser, and at -O3 WITH it when the
// <2 x float> operations it forms are kept away from the packed-FP32 instructions, -target-feature -packed-fp32-ops).
// tests/test_gpu_parity.py::test_merged_queries_in_lds_equal_the_golden pins the shipped object.
// -DMCPT_LEAN_POOL_ONLY_MERGED: experiment builds that bisect the compiler's passes on ONE kernel (tools/experiments/bisect_lean_merge.sh).
#define MCPT_UNIT_LEAN_POOL_MERGED
#include "render_kernel_impl.h"

namespace mcpt
{

#if defined(MCPT_LEAN_POOL_ONLY_MERGED)
template <> hipError_t Launch<kFeatEmitters | kPM, false, true>(MCPT_LAUNCH_ARGS) { return hipErrorNotSupported; }
template hipError_t Launch<kPM, false, true>(MCPT_LAUNCH_ARGS);
#else
template hipError_t Launch<kPM, false, true>(MCPT_LAUNCH_ARGS);
template hipError_t Launch<kFeatEmitters | kPM, false, true>(MCPT_LAUNCH_ARGS);
#endif

} // namespace mcpt
