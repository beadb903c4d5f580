// Who instantiates which render kernel of the LOW-DISCREPANCY build (throughput mode 2 of mcpt_renderer_set_rng): every unit
// that includes this header is compiled with MCPT_LOW_DISCREPANCY, i.e. a draw of the random stream returns an Owen-scrambled
// Sobol point (vecmath.h) — the production instantiations of the scene classes (hip/render_kernel.hip LaunchRender) once
// more, under names that carry kFeatLowDisc.  No reference counterpart (the reference's only low-discrepancy point is the
// radical inverse of the pixel jitter, math.hpp:29-41, which every mode keeps).
#ifndef MCPT_LOWDISC_UNITS_H
#define MCPT_LOWDISC_UNITS_H

#define MCPT_LOW_DISCREPANCY 1
#include "render_kernel_impl.h"

namespace mcpt
{

constexpr uint32_t kLD = kFeatLowDisc;

#if !defined(MCPT_UNIT_LOWDISC)
extern template hipError_t Launch<kAll | kLD, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kO | kLD, false, true>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kV | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_LOWDISC_2)
extern template hipError_t Launch<kP | kLD, false, true>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kFeatEmitters | kP | kLD, false, true>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kVolumeLean | kO | kLD, false, true>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_LOWDISC_3)
extern template hipError_t Launch<kFeatEmitters | kPB | kLD, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kFeatEmitters | kPB | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kSurface | kPB | kLD, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kSurface | kPB | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);
#endif
#if !defined(MCPT_UNIT_LOWDISC_4)
extern template hipError_t Launch<kAll | kPB | kLD, false, false>(MCPT_LAUNCH_ARGS);
extern template hipError_t Launch<kAll | kPB | kS | kLD, false, false>(MCPT_LAUNCH_ARGS);
#endif

} // namespace mcpt

#endif // MCPT_LOWDISC_UNITS_H
