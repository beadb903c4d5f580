// The kernel formulations the library's rule no longer chooses anywhere — the stream kernel (set_kernel 1 / 2 / 4), the queued renderer
// (5), the first multi-kernel version (3) and the lean trace-rate experiment — are built with `make EXPERIMENTAL=1` only (round 5;
// round 4's review, task 8).  Decided by one table: on the reference's eight loadable scenes at 640 x 360 spp 64 the lane-owns-a-path
// kernel with the pool walk is 1.4 - 4.2 x faster than the stream kernel in either of its forms (profiles/r05_rule_vs_calibrated.json:
// box 65 / 216 / 198 ms, classroom 110 / 178 / 155, dining-room 63 / 359 / 267, matpreview rough plastic 39 / 88 / 80, thin
// dielectric 73 / 122 / 122, dragon 18 / 67 / 65, rough conductor 43 / 86 / 76, rough dielectric 60 / 97 / 91); the queued renderer
// was 2 x slower than the stream kernel (EXPERIMENTS R3-1), mode 3 slower again.  This unit stands in for them in the default build:
// "not supported" everywhere, so that a request for one of those modes renders with the lane-owns-a-path kernel (what the library
// does for every scene a formulation does not cover) and the trace-rate experiment reports that it was not built.
#include <hip/hip_runtime.h>

#include "render_kernel_impl.h"

namespace mcpt
{

bool FormulationsBuilt() { return false; }

bool StreamSupports(const DeviceScene &, const RenderJob &) { return false; }
bool StreamPrefersLanes(const DeviceScene &sc) { return StagedBytes(sc, true) <= kLdsGeometryBytes; }
hipError_t PlanRenderStream(const DeviceScene &, const RenderJob &, bool, uint32_t, StreamLaunch *, const char **) { return hipErrorNotSupported; }
hipError_t LaunchRenderStream(const DeviceScene &, const RenderJob &, float *, TraceCounters *, hipStream_t, uint32_t *, const StreamLaunch &) { return hipErrorNotSupported; }

bool WavefrontSupports(const DeviceScene &, const RenderJob &) { return false; }
void WavefrontSizes(const DeviceScene &, uint32_t, size_t *cold_words, size_t *hot_words, size_t *id_words) { *cold_words = *hot_words = *id_words = 0; }
uint32_t WavefrontCounterWords() { return 0; }
hipError_t LaunchWavefrontRound(const DeviceScene &, const RenderJob &, float *, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t, bool, uint32_t, hipStream_t)
{
    return hipErrorNotSupported;
}

bool QueuedSupports(const DeviceScene &, const RenderJob &) { return false; }
uint32_t QueuedGroups(const BsdfRec *, size_t, bool) { return 0; }
void QueuedLayout(uint32_t, uint32_t, uint32_t, uint32_t, QueuedSizes *sizes) { *sizes = QueuedSizes{}; }
uint32_t QueuedTraceBlocks(uint32_t) { return 0; }
uint32_t *QueuedCounters(uint32_t *base, const QueuedSizes &) { return base; }
hipError_t LaunchQueuedRound(const DeviceScene &, const RenderJob &, float *, uint32_t *, const QueuedSizes &, uint32_t, uint32_t, uint32_t, hipStream_t) { return hipErrorNotSupported; }

hipError_t RunTraceRate(const DeviceScene &, uint32_t, const float *, int, int, uint32_t, uint32_t, uint32_t *, float *, hipStream_t) { return hipErrorNotSupported; }

} // namespace mcpt
