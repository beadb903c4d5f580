// TEST INFRASTRUCTURE — the 64-lane lockstep shim behind csrc/wave_target.h's MCPT_WAVE_EMU.
//
// The product's wavefront-cooperative code (csrc/pool_walk.h, the uniform / merged path steps of csrc/path_core.h, the kernel bodies
// render_body and sorted_body) is written for a gfx950 wavefront: 64 lanes that execute every instruction together, cross-lane
// operations (ballot, readfirstlane, mbcnt ranks), LDS as the lanes' shared memory, a fence + wave barrier where one lane reads what
// another wrote.  Until round 6 that code existed for the device only, and what pinned it were its OUTPUTS on the GPU.  Here the
// same source is compiled for the host:
//
//   * a workgroup is 256 FIBERS (ucontext) on one host thread, one per lane, each with its own stack;
//   * a lane runs until it reaches a cross-lane operation (a COLLECTIVE: ballot, readfirstlane, readlane, shuffle, the wavefront
//     barrier of pool_sync, __syncthreads, the top of the persistent loop) and parks there; when every lane of the workgroup is
//     parked (or has left the kernel) the scheduler completes, per wavefront, the collective of the lanes that stand DEEPEST in
//     the nesting of diverged regions — a block that only some lanes enter and that holds collectives starts with
//     MCPT_WAVE_REGION() (csrc/wave_target.h; nothing on the device): its lanes run, their ballots see each other only, while the
//     others wait behind it, which is what the execution mask does — with exactly those lanes as the active set.  Lanes of one
//     wavefront that stand at the same depth at DIFFERENT source positions are an error (a diverged block without its mark), so
//     the model checks its own assumption.  The top of the persistent loop (MCPT_WAVE_CONVERGE) completes last: a lane that
//     `continue`s waits there for the rest of its wavefront, like behind the hardware's loop latch;
//   * between two collectives the lanes run ONE AFTER THE OTHER, in ascending, descending or shuffled order: code that is correct
//     only because the hardware runs the lanes' instructions in lockstep — a read that must precede another lane's write with no
//     collective in between — renders a different frame under another order, and a protocol that holds under every order is
//     independent of it;
//   * LDS is a heap array of exactly the launch's dynamic size (AddressSanitizer sees an access beyond it; on the GPU such a write is
//     dropped and such a read returns 0, silently), and poison() fills a wavefront's pool area with a chosen pattern before every ray
//     query: two patterns, two frames — equal frames mean nothing is read before the query wrote it.
#ifndef MCPT_WAVE_SHIM_H
#define MCPT_WAVE_SHIM_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct uint2
{
    uint32_t x, y;
};

namespace wave_emu
{

enum Kind : uint32_t
{
    kBallot,
    kReadFirst,
    kReadLane,
    kShuffleXor,
    kWaveBarrier,
    kPoison,
    kConverge,
    kSleep,
    kSyncThreads,
};

// ---- the calling lane (a fiber of the running workgroup; outside a workgroup: a "wavefront" of one lane) ----
uint32_t thread_index();  // threadIdx.x
uint32_t block_index();   // blockIdx.x
uint32_t grid_blocks();   // gridDim.x
void *dynamic_lds_base(); // the workgroup's dynamic LDS
uint64_t collective(Kind kind, uint32_t site, uint64_t value, uint32_t aux = 0, void *ptr = nullptr);
uint32_t site_rank(const char *file, int line);
uint64_t tick();

template <class T>
inline T *dynamic_lds()
{
    return reinterpret_cast<T *>(dynamic_lds_base());
}

struct Coordinate
{
    uint32_t (*read)();
    operator uint32_t() const { return read(); }
};
struct Index3
{
    Coordinate x;
};
uint32_t block_size(); // blockDim.x: 256, or 128 in the class-sorted kernels

void region_enter();
void region_leave();
struct Region
{
    Region() { region_enter(); }
    ~Region() { region_leave(); }
    Region(const Region &) = delete;
};

inline void converge() { collective(kConverge, 0xFFFFFFF0u, 0); }
inline void poison(uint32_t *pool, uint32_t words) { collective(kPoison, 0xFFFFFFE0u, words, 0, pool); }

inline uint32_t mbcnt(uint32_t mask, uint32_t base, uint32_t lane_offset)
{
    // v_mbcnt_{lo,hi}_u32_b32: base + the number of set bits of the mask half below this lane
    const uint32_t lane = thread_index() & 63u;
    if (lane <= lane_offset)
        return base;
    const uint32_t below = lane - lane_offset;
    return base + static_cast<uint32_t>(__builtin_popcount(below >= 32u ? mask : (mask & ((1u << below) - 1u))));
}

} // namespace wave_emu

// ---- the HIP vocabulary the kernel bodies use ----
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)

static const ::wave_emu::Index3 threadIdx{{::wave_emu::thread_index}};
static const ::wave_emu::Index3 blockIdx{{::wave_emu::block_index}};
static const ::wave_emu::Index3 blockDim{{::wave_emu::block_size}};
static const ::wave_emu::Index3 gridDim{{::wave_emu::grid_blocks}};

#define MCPT_WAVE_SITE ::wave_emu::site_rank(__FILE__, __LINE__)

#define __lane_id() (::wave_emu::thread_index() & 63u)
#define __ballot(p) ::wave_emu::collective(::wave_emu::kBallot, MCPT_WAVE_SITE, (p) ? 1u : 0u)
#define __popcll(x) __builtin_popcountll(x)
#define __ffsll(x) __builtin_ffsll(x)
#define __builtin_amdgcn_readfirstlane(v) static_cast<int>(::wave_emu::collective(::wave_emu::kReadFirst, MCPT_WAVE_SITE, static_cast<uint32_t>(v)))
#define __builtin_amdgcn_readlane(v, l) static_cast<uint32_t>(::wave_emu::collective(::wave_emu::kReadLane, MCPT_WAVE_SITE, static_cast<uint32_t>(v), static_cast<uint32_t>(l)))
#define __shfl_xor(v, off, width) static_cast<uint32_t>(::wave_emu::collective(::wave_emu::kShuffleXor, MCPT_WAVE_SITE, static_cast<uint32_t>(v), static_cast<uint32_t>(off)))
#define __builtin_amdgcn_mbcnt_lo(mask, base) ::wave_emu::mbcnt((mask), (base), 0u)
#define __builtin_amdgcn_mbcnt_hi(mask, base) ::wave_emu::mbcnt((mask), (base), 32u)
// (the fence orders the compiler's accesses; the barrier that follows it is where the lanes meet)
#define __builtin_amdgcn_fence(order, scope) __atomic_signal_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_wave_barrier() static_cast<void>(::wave_emu::collective(::wave_emu::kWaveBarrier, MCPT_WAVE_SITE, 0))
#define __builtin_amdgcn_s_sleep(n) static_cast<void>(::wave_emu::collective(::wave_emu::kSleep, 0xFFFFFFF8u, 0))
#define __syncthreads() static_cast<void>(::wave_emu::collective(::wave_emu::kSyncThreads, MCPT_WAVE_SITE, 0))
// (real acquire / release on the host threads that run the workgroups: the path market's records travel between them)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define __threadfence() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_getreg(immediate) (::wave_emu::block_index() & 7u) /* HW_REG_XCC_ID: workgroup b runs on XCD b % 8 */
#define __HIP_MEMORY_SCOPE_AGENT 0

inline uint32_t __float_as_uint(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline float __uint_as_float(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// (workgroups run on several host threads: device-memory atomics are real ones; LDS belongs to one thread)
inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline uint32_t atomicCAS(uint32_t *p, uint32_t expected, uint32_t desired)
{
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return expected; // (the old value, like the device's)
}
inline uint32_t atomicMin(uint32_t *p, uint32_t v)
{
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
    {
    }
    return old;
}
inline unsigned long long clock64() { return ::wave_emu::tick(); }
inline unsigned long long wall_clock64() { return ::wave_emu::tick(); }

#endif // MCPT_WAVE_SHIM_H
