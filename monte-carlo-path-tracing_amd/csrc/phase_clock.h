// Diagnostic of experiment builds: where a wavefront's time goes inside a step (tools/experiments/phase_clock.py).
#ifndef MCPT_PHASE_CLOCK_H
#define MCPT_PHASE_CLOCK_H

#include "vecmath.h"

namespace mcpt
{

// ---- diagnostic, experiment builds only (-DMCPT_PHASE_CLOCK=1 on hip/sorted_kernel.hip; tools/experiments/phase_clock.py) ----
// Where a wavefront's time goes inside a step.  phase_mark(p) books the shader-clock cycles since the wavefront's previous mark
// to phase p, together with the number of lanes that arrive at the mark (the exec mask there).  One lane per wavefront does the
// booking, into that wavefront's words of LDS; the kernel adds them up into RenderJob::wave_clock when it ends.  Compiled out
// (an empty function) everywhere else.
#ifndef MCPT_PHASE_CLOCK
#define MCPT_PHASE_CLOCK 0
#endif
enum : uint32_t { kPhaseRegenerate, kPhaseExtend, kPhaseSurface, kPhaseMedium, kPhaseResolve, kPhaseSort, kPhaseEmitter, kPhaseShadow, kPhaseWeigh,
                  kPhaseAreaLight, kPhasePhase, kPhaseBsdf, kPhaseScatter, kPhaseTriangleFrame, kPhaseQuadricFrame, kPhaseCount };
#if MCPT_PHASE_CLOCK && defined(__HIPCC__)
__device__ __forceinline__ unsigned long long *phase_area()
{
    __shared__ unsigned long long area[4][1 + 3 * kPhaseCount]; // per wavefront: last mark, then cycles / visits / lanes per phase
    return area[threadIdx.x >> 6];
}
__device__ __forceinline__ void phase_mark(uint32_t p, bool counted = true)
{
    const unsigned long long here = __ballot(true), lanes = __ballot(counted);
    if (__lane_id() == static_cast<uint32_t>(__ffsll(static_cast<long long>(here))) - 1u)
    {
        unsigned long long *a = phase_area();
        const unsigned long long now = clock64();
        a[1 + p] += now - a[0], a[1 + kPhaseCount + p] += 1, a[1 + 2 * kPhaseCount + p] += static_cast<unsigned long long>(__popcll(lanes));
        a[0] = clock64();
    }
}
#else
MCPT_HD void phase_mark(uint32_t, bool = true) {}
#endif

} // namespace mcpt

#endif // MCPT_PHASE_CLOCK_H
