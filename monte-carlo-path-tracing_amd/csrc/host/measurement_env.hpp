// MEASUREMENT SWITCHES (environment variables that change how a shipped library schedules its work: MCPT_LEVELS, MCPT_COST_ORDER,
// MCPT_POOL_ORDER, MCPT_TREELET, MCPT_WAVE_CLOCK, ...) exist in EXPERIMENT builds only: `make EXTRA=-DMCPT_MEASUREMENT_HOOKS=1` /
// tools/experiments/build_exp2.sh.  The default library reads none of them ("What's weak" 11: 25 such hooks
// silently changed scheduling); what it does read from the environment is part of its interface and documented in include/mcpt.h:
// MCPT_CHECK_WALKS, MCPT_CALIBRATE / MCPT_CALIBRATION_FILE, MCPT_RCCL_LIBRARY, MCPT_MESH_TANGENTS.
#ifndef MCPT_MEASUREMENT_ENV_HPP
#define MCPT_MEASUREMENT_ENV_HPP

#include <cstdlib>

#ifndef MCPT_MEASUREMENT_HOOKS
#define MCPT_MEASUREMENT_HOOKS 0
#endif

namespace mcpt
{

inline const char *MeasurementEnv(const char *name)
{
#if MCPT_MEASUREMENT_HOOKS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

} // namespace mcpt

#endif // MCPT_MEASUREMENT_ENV_HPP
