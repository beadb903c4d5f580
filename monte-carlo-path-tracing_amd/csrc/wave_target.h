// Which target the wavefront-cooperative code (pool_walk.h, the uniform path steps of path_core.h, the kernel bodies of
// hip/render_kernel_impl.h and hip/sorted_body.h) is compiled for.
//
//   hipcc, device pass   the product: gfx950 code
//   hipcc, host pass     declarations only (the kernels' launchers)
//   MCPT_WAVE_EMU        TEST INFRASTRUCTURE (tests/emu/wave_shim.h, SURVEY section 5: "test the compaction / sort stages against a
//                        serial host model"): the SAME source on the host, a workgroup = 256 fibers stepped in lockstep between the
//                        cross-lane operations (ballot, readfirstlane, mbcnt, the wavefront-scope fence + barrier of pool_sync,
//                        __syncthreads), LDS a plain array with its launch size — under AddressSanitizer / UBSan and with poisoned
//                        pool areas in tests/test_wave_emu.py.  Never part of libmcpt_hip.so.
//   any other host build the per-lane functions only (host/host_render.cpp, tests/emu/emulator.cpp): a "wavefront" is one lane
#ifndef MCPT_WAVE_TARGET_H
#define MCPT_WAVE_TARGET_H

#if defined(MCPT_WAVE_EMU)
#include "wave_shim.h" // (tests/emu, on the include path of that build only)
#define MCPT_WAVE_CODE 1
#define MCPT_WAVE_DEVICE 1
// the kernel's dynamic LDS: an array of exactly the launch's size (an access beyond it is an error there, not a dropped write)
#define MCPT_DYNAMIC_LDS(type, name) type *name = ::wave_emu::dynamic_lds<type>()
// the top of a kernel's persistent loop: where the hardware's wavefront is whole again after a `continue`
#define MCPT_WAVE_CONVERGE() ::wave_emu::converge()
// the first statement of a block that only SOME lanes of a wavefront enter and that holds cross-lane operations: those lanes run
// it (their ballots see each other only) while the others wait behind it — what the execution mask does on the device
#define MCPT_WAVE_REGION() ::wave_emu::Region mcpt_wave_region_guard
// a ray query starts: nothing of what the pool area holds may be read before the query writes it
#define MCPT_POOL_POISON(pool, words) ::wave_emu::poison(pool, words)
#elif defined(__HIPCC__)
#define MCPT_WAVE_CODE 1
#if defined(__HIP_DEVICE_COMPILE__)
#define MCPT_WAVE_DEVICE 1
#else
#define MCPT_WAVE_DEVICE 0
#endif
#define MCPT_DYNAMIC_LDS(type, name) extern __shared__ type name[]
#define MCPT_WAVE_CONVERGE()
#define MCPT_WAVE_REGION()
// -DMCPT_POOL_POISON_WORD=<pattern>: experiment builds fill a wavefront's pool area with the pattern before every query (two
// patterns, two frames: equal frames = nothing is read before it is written).  Off in the product.
#if defined(MCPT_POOL_POISON_WORD)
#define MCPT_POOL_POISON(pool, words)                                                        \
    do                                                                                       \
    {                                                                                        \
        for (uint32_t poison_i = __lane_id(); poison_i < (words); poison_i += 64u)           \
            (pool)[poison_i] = (MCPT_POOL_POISON_WORD);                                      \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                               \
        __builtin_amdgcn_wave_barrier();                                                     \
    } while (0)
#else
#define MCPT_POOL_POISON(pool, words)
#endif
#else
#define MCPT_WAVE_CODE 0
#define MCPT_WAVE_DEVICE 0
#define MCPT_WAVE_REGION()
#endif

#endif // MCPT_WAVE_TARGET_H
