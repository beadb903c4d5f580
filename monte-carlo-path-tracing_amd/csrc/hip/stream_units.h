// Which translation unit instantiates which stream-kernel variant (stream_variants.inc): a unit defines
// MCPT_STREAM_UNIT (0..8) before including this header and gets explicit instantiations of its own variants;
// every other unit's variants are declared `extern template`, so that `make -j` compiles the units side by side.
#ifndef MCPT_STREAM_UNITS_H
#define MCPT_STREAM_UNITS_H

#include "render_kernel_impl.h"
#include "stream_kernel_impl.h"

namespace mcpt
{

#define MCPT_STREAM_PLAN_ARGS const DeviceScene &, const RenderJob &, uint32_t, StreamLaunch &
#define MCPT_STREAM_LAUNCH_ARGS \
    const DeviceScene &, const RenderJob &, float *, TraceCounters *, hipStream_t, uint32_t *, const StreamLaunch &
#define MCPT_STREAM_EXTERN(f, S, c, s, h, r)                                                \
    extern template hipError_t PlanStream<(f), S, c, s, h, r>(MCPT_STREAM_PLAN_ARGS);       \
    extern template hipError_t LaunchStream<(f), S, c, s, h, r>(MCPT_STREAM_LAUNCH_ARGS);
#define MCPT_STREAM_DEFINE(f, S, c, s, h, r)                                         \
    template hipError_t PlanStream<(f), S, c, s, h, r>(MCPT_STREAM_PLAN_ARGS);       \
    template hipError_t LaunchStream<(f), S, c, s, h, r>(MCPT_STREAM_LAUNCH_ARGS);

#if MCPT_STREAM_UNIT == 0
#define MCPT_STREAM_DECL_0 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_0 MCPT_STREAM_EXTERN
#endif
#if MCPT_STREAM_UNIT == 1
#define MCPT_STREAM_DECL_1 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_1 MCPT_STREAM_EXTERN
#endif
#if MCPT_STREAM_UNIT == 2
#define MCPT_STREAM_DECL_2 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_2 MCPT_STREAM_EXTERN
#endif
#if MCPT_STREAM_UNIT == 3
#define MCPT_STREAM_DECL_3 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_3 MCPT_STREAM_EXTERN
#endif

#if MCPT_STREAM_UNIT == 4
#define MCPT_STREAM_DECL_4 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_4 MCPT_STREAM_EXTERN
#endif
#if MCPT_STREAM_UNIT == 5
#define MCPT_STREAM_DECL_5 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_5 MCPT_STREAM_EXTERN
#endif

#if MCPT_STREAM_UNIT == 6
#define MCPT_STREAM_DECL_6 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_6 MCPT_STREAM_EXTERN
#endif
#if MCPT_STREAM_UNIT == 7
#define MCPT_STREAM_DECL_7 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_7 MCPT_STREAM_EXTERN
#endif

#if MCPT_STREAM_UNIT == 8
#define MCPT_STREAM_DECL_8 MCPT_STREAM_DEFINE
#else
#define MCPT_STREAM_DECL_8 MCPT_STREAM_EXTERN
#endif

#define X(index, features, S, counted, small, hot, regs, unit, name) MCPT_STREAM_DECL_##unit(features, S, counted, small, hot, regs)
#include "stream_variants.inc"
#undef X

} // namespace mcpt

#endif // MCPT_STREAM_UNITS_H
