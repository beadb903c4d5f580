// Shade launch of material group 5 of the queued renderer (queued_kernels.h): one translation unit per group, so that
// the seven instantiations — each with one BSDF model compiled in — build side by side.
#include "queued_kernels.h"

namespace mcpt
{
hipError_t LaunchQueuedShade5(const DeviceScene &sc, const RenderJob &job, float *out, const QueueView &qv, uint32_t parity, bool fresh,
                              uint32_t n_cus, hipStream_t stream)
{
    return LaunchQueuedShadeGroup<5>(sc, job, out, qv, parity, fresh, n_cus, stream);
}
} // namespace mcpt
